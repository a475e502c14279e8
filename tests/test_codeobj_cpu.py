"""The built gfx950 code objects, read back without a GPU: the MFMA kernels must keep their accumulators in registers.  (A conditional
load added to a fully unrolled epilogue once made hipcc place the accumulator arrays of the 256 x 128 bf16 kernels in scratch memory --
576 bytes per lane, 44 % slower, every parity test still green.)"""
import os

import pytest

from tests.codeobj_util import kernels

SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'patchgan_amd', 'libpatchgan_hip.so')
HOT = ('k_conv_bf16x', 'k_wgrad_bf16x', 'k_wino_gemm', 'k_wino_bgemm', 'k_wino_wgrad_gemm', 'k_b2s_fast', 'k_s2b_fast', 'k_wgrad_fast',
       'k_b2s_bf16', 'k_s2b_bf16', 'k_wgrad_bf16', 'k_big2small', 'k_small2big', 'k_wgrad')


@pytest.mark.skipif(not os.path.exists(SO), reason='library not built')
def test_mfma_kernels_keep_accumulators_in_registers():
    ks = kernels(SO)
    assert len(ks) > 100, len(ks)
    hot = {n: k for n, k in ks.items() if any(h in n for h in HOT)}
    assert len(hot) > 60
    bad = {}
    for n, k in hot.items():
        # a few dwords of spill in the densest kernels are tolerated; an accumulator array in scratch is hundreds of bytes.
        # (the MUL instantiation of the 128-tile-row stride-1 Winograd kernel spills 336 bytes in its epilogue: not on any benchmarked path)
        limit = 400 if ('k_wino_gemm' in n and 'Lb1E' in n) else 64
        if k['.private_segment_fixed_size'] > limit:
            bad[n] = k['.private_segment_fixed_size']
    assert not bad, bad
    # the bf16 LDS-DMA kernels run two workgroups per CU: <= 256 registers per lane
    for n, k in hot.items():
        if 'k_conv_bf16x' in n or 'k_wgrad_bf16x' in n:
            assert k['.vgpr_count'] + k.get('.agpr_count', 0) <= 256, (n, k['.vgpr_count'])
