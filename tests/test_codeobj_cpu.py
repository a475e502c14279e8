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


OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'


@pytest.mark.skipif(not os.path.exists(SO), reason='library not built')
def test_no_packed_f32_low_half_from_the_odd_register_of_a_fresh_lds_pair(tmp_path):
    """Cause of round 3's open issue (k_s2b_ca1_s1 returning different values next to a second process on the GPU): `acc += x * w`
    with x read from LDS compiled to v_pk_mul_f32 ... op_sel:[0,1] straight off a ds_read2_b32 pair; lanes 48-63 of the low half
    then used the register's old content although s_waitcnt lgkmcnt was satisfied (one tap of the sum dropped; tools/forensics/debug_cc5.py,
    two processes: an event in 50 of 92 repetitions of 18 steps; with the values passed through v_readfirstlane / v_mov_b32 first:
    0 of 92).  No kernel of the library may contain that operand form.  The guard must not pass vacuously: with a built library and no
    disassembler the test FAILS (validated against hipcc / AMD clang 22.0.0git of ROCm 7.2.0, HIP 7.2.26015)."""
    assert os.path.exists(OBJDUMP), f'{OBJDUMP} missing: the ISA scan of the built library cannot run'
    import subprocess
    from tests.codeobj_util import code_objects, packed_f32_reads_of_fresh_lds_pairs
    objs = code_objects(SO)
    assert len(objs) >= 7
    bad = {}
    ninstr = 0
    for i, co in enumerate(objs):
        elf = tmp_path / f'co{i}.elf'
        elf.write_bytes(co)
        dis = subprocess.run([OBJDUMP, '-d', '--mcpu=gfx950', str(elf)], check=True, capture_output=True, text=True).stdout.splitlines()
        ninstr += len(dis)
        for fn, ins in packed_f32_reads_of_fresh_lds_pairs(dis):
            bad.setdefault(fn, []).append(ins)
        del dis
    assert ninstr > 500000, ninstr
    assert not bad, {k: v[:2] for k, v in bad.items()}


def test_the_scan_recognises_the_failing_form():
    from tests.codeobj_util import packed_f32_reads_of_fresh_lds_pairs
    bad = ['0000000000001000 <k>:',
           '\tds_read2_b32 v[92:93], v0 offset0:6 offset1:7          // 000000001000: D86E0706',
           '\ts_waitcnt lgkmcnt(0)                                  // 000000001008: BF8CC07F',
           '\tv_pk_mul_f32 v[0:1], v[12:13], v[92:93] op_sel:[0,1]   // 00000000100C: D3B14000']
    assert [i for _, i in packed_f32_reads_of_fresh_lds_pairs(bad)] == ['v_pk_mul_f32 v[0:1], v[12:13], v[92:93] op_sel:[0,1]']
    ok = bad[:3] + ['\tv_mov_b32_e32 v93, v93', bad[3]]
    assert not list(packed_f32_reads_of_fresh_lds_pairs(ok))
    natural = bad[:3] + ['\tv_pk_mul_f32 v[0:1], v[12:13], v[92:93] op_sel_hi:[1,0]']       # high half from the even register: never seen to fail
    assert not list(packed_f32_reads_of_fresh_lds_pairs(natural))


@pytest.mark.skipif(not os.path.exists(SO), reason='library not built')
def test_hot_kernels_keep_the_occupancy_they_were_tuned_at():
    """Round 4's two speed-ups were occupancy accidents, not algorithms: the bf16 LDS-DMA tiles with 64 accumulators ran two workgroups
    per CU where four fit (cfg4 bf16 step 6.94 -> 6.35 ms), and hipcc had built the multi-batch fp32 Winograd GEMM at 196 VGPRs + 64
    AGPRs = ONE wave per SIMD (cfg2 step 9.07 -> 8.90 ms with the four-wave kernel).  Waves per SIMD = min(8, 512 // allocated
    registers) with an allocation granule of 8 (MI355X_MICROARCH.md, register files); a change that pushes one of these kernels over its
    budget costs that much again, silently."""
    ks = kernels(SO)

    def waves(k):
        regs = k['.vgpr_count'] + k.get('.agpr_count', 0)
        return min(8, 512 // (((regs + 7) // 8) * 8))

    want = {  # substring of the mangled name -> minimum waves per SIMD by registers
        'k_wino_bgemmILi2ELi2ELi2ELi2E': 4, 'k_wino_bgemmILi1ELi2ELi2ELi2E': 4, 'k_wino_wgrad_gemmILi1ELi1ELi2ELi2E': 4,
        # the split-bf16 batched GEMM (round 6): 128-row tiles two workgroups per CU (57 KB of LDS each), 64-row tiles three
        'k_wino_bgemm_s3ILi2ELi2ELi2ELi2ELi2E': 2, 'k_wino_bgemm_s3ILi1ELi2ELi2ELi2ELi3E': 3,
        'k_wino_wgrad_gemm_s3ILi2ELi2ELi2ELi2ELi1ELi2E': 2, 'k_wino_wgrad_gemm_s3ILi1ELi1ELi2ELi2ELi2ELi3E': 3, 'k_wino_gemm_row_s3ILi2E': 2,
        'k_wino_wgrad_gemmILi2ELi2ELi2ELi2E': 4, 'k_wino_gemm_rowILi4ELi1E': 4, 'k_wgrad_fastILi2ELi2ELi2ELi2E': 4,
        'k_b2s_fastILi2ELi2ELi2ELi2E': 3, 'k_s2b_fastILi2ELi2ELi2ELi2E': 3,
    }
    seen = {w: 0 for w in want}
    seen['bf16 OCC4'] = 0
    bad = {}
    for n, k in ks.items():
        for w, need in want.items():
            if w in n:
                seen[w] += 1
                if waves(k) < need or k['.private_segment_fixed_size'] > 64:
                    bad[n] = (waves(k), need, k['.private_segment_fixed_size'])
        # the 64-accumulator bf16 tiles at OCC 4 (trailing template argument 4): <= 128 registers, no scratch, LDS for four per CU
        if ('k_conv_bf16x' in n or 'k_wgrad_bf16x' in n) and 'ELi4EEEv' in n:
            wgs = 16 // (k['.max_flat_workgroup_size'] // 64)          # workgroups per CU at four waves per SIMD (four- or eight-wave tiles)
            if waves(k) < 4 or k['.private_segment_fixed_size'] > 0 or wgs * k['.group_segment_fixed_size'] > 160 * 1024:
                bad[n] = (waves(k), 4, k['.private_segment_fixed_size'], k['.group_segment_fixed_size'])
            seen['bf16 OCC4'] = seen.get('bf16 OCC4', 0) + 1
        # the window-staged bf16 kernel: two workgroups per CU -- <= 256 registers, NO scratch (its main loop keeps four fragment sets
        # and 64 accumulators live: a spill lands in the loop), two workgroups' LDS within the CU's 160 KB
        if 'k_conv_bf16r' in n:
            if waves(k) < 2 or k['.private_segment_fixed_size'] > 0 or 2 * k['.group_segment_fixed_size'] > 160 * 1024:
                bad[n] = (waves(k), 2, k['.private_segment_fixed_size'], k['.group_segment_fixed_size'])
            seen['bf16 window'] = seen.get('bf16 window', 0) + 1
    assert all(v > 0 for v in seen.values()), seen
    assert seen['bf16 OCC4'] >= 10 and seen.get('bf16 window', 0) >= 10, seen
    assert not bad, bad
