"""Seeded random geometries through the three conv entry points: PG_ALGO_AUTO (implicit GEMM + every Winograd / small-channel
path the planner picks) against PG_ALGO_DIRECT (one thread per output, no tiling) of the same library, on the GPU only.
Covers ragged tiles, odd extents, N = 1, channel counts around the eligibility thresholds; every seed runs with the default
heuristics and with the polyphase / F(3x3,4x4) paths forced wherever the geometry allows (per-call PG_TUNE_* bits)."""
import math
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


def _geoms(n, seed):
    rng = random.Random(seed)
    out = []
    while len(out) < n:
        s = rng.choice([1, 2, 2])
        N = rng.choice([1, 2, 3, 5, 8, 16])
        Hb = rng.randint(6, 72)
        Wb = rng.randint(6, 72)
        Ca = rng.choice([1, 3, 4, 8, 24, 32, 64, 96, 128, 160, 256, 288])
        Cb = rng.choice([1, 3, 4, 8, 24, 32, 40, 64, 96, 128, 160, 256])
        if N * Hb * Wb * max(Ca, Cb) > 6_000_000 or Ca * Cb > 40_000:
            continue
        out.append((N, Hb, Wb, Ca, Cb, s))
    return out


import os


@pytest.mark.parametrize('tuning', ['default', 'forced'])
@pytest.mark.parametrize('seed', [0, 1, 2] + ([int(v) for v in os.environ['PATCHGAN_FUZZ_SEEDS'].split(',')] if os.environ.get('PATCHGAN_FUZZ_SEEDS') else []))
def test_auto_matches_direct_on_random_geometries(seed, tuning):
    from patchgan_amd import engine as E, _lib as L
    from tests.gpu_util import to_view, empty_view, pack, unpack, rel_err, DEV
    bits = 0 if tuning == 'default' else (L.TUNE_WINO2_ALL | L.TUNE_WINO2W_ALL | L.TUNE_WINO1_F3)
    picked = set()
    for geom in _geoms(40, seed):
        N, Hb, Wb, Ca, Cb, s = geom
        g = torch.Generator().manual_seed(hash(geom) & 0xFFFF)
        Hs, Ws = (Hb - 2) // s + 1, (Wb - 2) // s + 1
        big = torch.randn(N, Cb, Hb, Wb, generator=g)
        small = torch.randn(N, Ca, Hs, Ws, generator=g)
        Wt = torch.randn(Ca, Cb, 4, 4, generator=g) / math.sqrt(max(Ca, Cb) * 16)     # outputs of order 1 in both directions
        ba, bb = torch.randn(Ca, generator=g).cuda(), torch.randn(Cb, generator=g).cuda()
        P = pack(Wt)
        res = {}
        for algo in (0, 1):
            op = E.ConvOp(*geom, algo | (bits if algo == 0 else 0))
            if algo == 0:
                picked.update(op.describe(i)[0].split('<')[0] for i in range(3))
            vs = empty_view(N, Hs, Ws, Ca, ld=Ca + 4, off=4)
            op.big2small(to_view(big, ld=Cb + 4), P, 0, ba, 0, vs, 1)
            vb = empty_view(N, Hb, Wb, Cb, ld=Cb + 8, off=4)
            op.small2big(to_view(small, ld=Ca + 4, off=4), P, 0, bb, 0, vb, 1)
            dP = torch.full((16 * Ca * Cb,), float('nan'), device=DEV)
            db = torch.full((Ca,), float('nan'), device=DEV)
            op.wgrad(to_view(small, ld=Ca + 4, off=4), to_view(big, ld=Cb + 8, off=4), dP, 0, db, 0)
            torch.cuda.synchronize()
            res[algo] = (vs.to_nchw(), vb.to_nchw(), unpack(dP, Ca, Cb), db.clone())
        assert rel_err(res[0][0], res[1][0]) < 2e-5, ('big2small', geom)
        assert rel_err(res[0][1], res[1][1]) < 2e-5, ('small2big', geom)
        assert rel_err(res[0][2], res[1][2]) < 3e-5, ('wgrad', geom)
        assert rel_err(res[0][3], res[1][3]) < 3e-5, ('dbias', geom)
    print('kernel families exercised:', sorted(picked))


def _geoms_bf16(n, seed):
    rng = random.Random(1000 + seed)
    out = []
    while len(out) < n:
        s = rng.choice([1, 2, 2])
        N = rng.choice([1, 2, 3, 5, 8])
        Hb = rng.randint(4, 70)
        Wb = rng.randint(4, 70)
        Ca = rng.choice([32, 64, 64, 96, 128, 192, 256, 320])
        Cb = rng.choice([1, 3, 4, 7, 8, 64, 64, 128, 192, 256])
        if N * Hb * Wb * max(Ca, Cb) > 5_000_000 or Ca * Cb > 50_000:
            continue
        out.append((N, Hb, Wb, Ca, Cb, s))
    return out


@pytest.mark.parametrize('staging', ['flat', 'ring'])
@pytest.mark.parametrize('seed', [0, 1])
def test_bf16_lds_dma_kernels_match_direct_on_random_geometries(seed, staging):
    """The bf16-tensor kernels of conv_bf16.hip (k_conv_bf16x dir 0..3, k_wgrad_bf16x incl. taps-in-N, split-K, ragged tiles, odd
    parity classes, stride 1, 8-channel-pixel image-facing tensors) against PG_ALGO_DIRECT (fp32, one thread per output) on
    bf16-representable operands: every product is exact, so fp32 results agree to summation order (2e-5) and bf16 results to one
    bf16 ulp of the rounded direct result."""
    from patchgan_amd import engine as E, _lib as L
    from tests.gpu_util import to_view, to_view_bf, to_view_bf8, empty_view, empty_view_bf, pack, unpack, rel_err, DEV
    tune = L.TUNE_BF16X_RING if staging == 'ring' else L.TUNE_BF16X_FLAT
    picked = set()
    for geom in _geoms_bf16(30, seed):
        N, Hb, Wb, Ca, Cb, s = geom
        g = torch.Generator().manual_seed(hash(geom) & 0xFFFF)
        Hs, Ws = (Hb - 2) // s + 1, (Wb - 2) // s + 1
        big = torch.randn(N, Cb, Hb, Wb, generator=g).bfloat16().float()
        small = torch.randn(N, Ca, Hs, Ws, generator=g).bfloat16().float()
        Wt = (torch.randn(Ca, Cb, 4, 4, generator=g) / math.sqrt(max(Ca, Cb) * 16)).bfloat16().float()
        P = pack(Wt)
        ref, op = E.ConvOp(*geom, L.ALGO_DIRECT), E.ConvOp(*geom, L.ALGO_BF16 | tune)
        few = Cb <= 8
        vb_bf = to_view_bf8(big) if few else to_view_bf(big, ld=Cb + 8, off=8)
        vs_bf = to_view_bf(small, ld=Ca + 8, off=8)

        def close_bf(got, want):
            wr = want.float().bfloat16().double().cpu()
            return ((got.double().cpu() - wr).abs() <= wr.abs() * 2.0 ** -7 + 1e-5 * wr.abs().max()).all()
        # big -> small
        if few or Cb % 64 == 0:
            want = empty_view(N, Hs, Ws, Ca)
            ref.big2small(to_view(big), P, 0, None, 0, want, 1)
            got = empty_view_bf(N, Hs, Ws, Ca, ld=Ca + 8, off=8)
            op.big2small(vb_bf, P, 0, None, 0, got, 1)
            torch.cuda.synchronize()
            picked.add(op.describe(0, L.IO_MASK)[0])
            assert close_bf(got.to_nchw(), want.to_nchw()), ('big2small', geom)
        # small -> big (fp32 result onto few channels, bf16 otherwise)
        if Ca % 64 == 0:
            want = empty_view(N, Hb, Wb, Cb)
            ref.small2big(to_view(small), P, 0, None, 0, want, 1)
            got = (empty_view if few else empty_view_bf)(N, Hb, Wb, Cb, ld=Cb + 8, off=8)
            op.small2big(vs_bf, P, 0, None, 0, got, 1)
            torch.cuda.synchronize()
            picked.add(op.describe(1, L.IO_SMALL_BF16 if few else L.IO_MASK)[0])
            if few:
                assert rel_err(got.to_nchw(), want.to_nchw()) < 2e-5, ('small2big', geom)
            else:
                assert close_bf(got.to_nchw(), want.to_nchw()), ('small2big', geom)
        # weight gradient
        if Ca % 32 == 0 and Ca >= 64 and (few or Cb % 32 == 0):
            dPr = torch.full((16 * Ca * Cb,), float('nan'), device=DEV)
            dPg = torch.full((16 * Ca * Cb,), float('nan'), device=DEV)
            ref.wgrad(to_view(small), to_view(big), dPr, 0)
            op.wgrad(vs_bf, vb_bf, dPg, 0)
            torch.cuda.synchronize()
            picked.add(op.describe(2, L.IO_MASK)[0])
            assert rel_err(dPg, dPr) < 3e-5, ('wgrad', geom)
    print('bf16 kernels exercised:', sorted(picked))
    assert sum(k.startswith('k_conv_bf16x') for k in picked) >= 6 and sum(k.startswith('k_wgrad_bf16x') for k in picked) >= 3, sorted(picked)
