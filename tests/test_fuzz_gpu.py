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
    assert sum(k.startswith('k_conv_bf16x') or k.startswith('k_conv_bf16r') for k in picked) >= 6 and sum(k.startswith('k_wgrad_bf16x') for k in picked) >= 3, sorted(picked)


def _geoms_window(n, seed):
    """Geometries k_conv_bf16r takes in at least one direction: stride 2, even maps, class maps of whole R x 16 rectangles."""
    rng = random.Random(2000 + seed)
    out = []
    while len(out) < n:
        N = rng.choice([1, 2, 3])
        Hs = 8 * rng.randint(1, 6)
        Ws = 16 * rng.randint(1, 3)
        Ca = rng.choice([64, 64, 128, 192, 256, 512])
        Cb = rng.choice([64, 64, 128, 192, 320, 512])
        if N * 4 * Hs * Ws * max(Ca, Cb) > 6_000_000 or Ca * Cb > 140_000:
            continue
        out.append((N, 2 * Hs, 2 * Ws, Ca, Cb, 2))
    return out


@pytest.mark.parametrize('seed', [0, 1])
def test_bf16_window_kernel_matches_direct_on_random_geometries(seed):
    """k_conv_bf16r (window-staged stride-2 kernel, both directions, both tilings, K splits, ragged channel tiles, fp32 and bf16
    outputs, the activation-derivative multiplier and the InstanceNorm partial sums in its epilogue) against PG_ALGO_DIRECT on
    bf16-representable operands: products exact, so fp32 results agree to summation order and bf16 results to one ulp of the rounded
    direct result; the partial sums are those of the stored tensor."""
    from patchgan_amd import engine as E, _lib as L
    from tests.gpu_util import to_view, to_view_bf, empty_view, empty_view_bf, pack, rel_err, DEV
    ran = {0: 0, 1: 0}
    extras = {'mul': 0, 'part': 0, 'split': 0}
    for geom in _geoms_window(14, seed):
        N, Hb, Wb, Ca, Cb, s = geom
        g = torch.Generator().manual_seed(hash(geom) & 0xFFFF)
        Hs, Ws = Hb // 2, Wb // 2
        big = torch.randn(N, Cb, Hb, Wb, generator=g).bfloat16().float()
        small = torch.randn(N, Ca, Hs, Ws, generator=g).bfloat16().float()
        Wt = (torch.randn(Ca, Cb, 4, 4, generator=g) / math.sqrt(max(Ca, Cb) * 16)).bfloat16().float()
        bias = torch.randn(max(Ca, Cb), generator=g)
        P = pack(Wt)
        ref, op = E.ConvOp(*geom, L.ALGO_DIRECT), E.ConvOp(*geom, L.ALGO_BF16)

        def close_bf(got, want):
            wr = want.float().bfloat16().double().cpu()
            return ((got.double().cpu() - wr).abs() <= wr.abs() * 2.0 ** -7 + 1e-5 * wr.abs().max()).all()

        for opcode in (0, 1):
            sym, split = op.describe(opcode, L.IO_MASK)
            if not sym.startswith('k_conv_bf16r'):
                continue
            ran[opcode] += 1
            extras['split'] += split > 1
            cout = Ca if opcode == 0 else Cb
            oshape = (N, Hs, Ws, Ca) if opcode == 0 else (N, Hb, Wb, Cb)
            src_f = to_view(big if opcode == 0 else small)
            src = to_view_bf(big if opcode == 0 else small, ld=(Cb if opcode == 0 else Ca) + 8, off=8)
            call_r = ref.big2small if opcode == 0 else ref.small2big
            call = op.big2small if opcode == 0 else op.small2big
            want = empty_view(*oshape)
            call_r(src_f, P, 0, bias.cuda(), 0, want, 1)
            for out_bf in (True, False):
                got = (empty_view_bf if out_bf else empty_view)(*oshape, ld=cout + 8, off=8)
                call(src, P, 0, bias.cuda(), 0, got, 1)
                torch.cuda.synchronize()
                if out_bf:
                    assert close_bf(got.to_nchw(), want.to_nchw()), (geom, opcode, sym, split)
                else:
                    assert rel_err(got.to_nchw(), want.to_nchw()) < 2e-5, (geom, opcode, sym, split)
            # InstanceNorm partial sums from the epilogue (bf16 output, no bias / activation)
            y = empty_view_bf(*oshape, ld=cout + 8, off=8)
            chunks = op.stats_chunks(opcode, src, y)
            if chunks:
                extras['part'] += 1
                part = torch.full((N * chunks * cout * 2,), float('nan'), dtype=torch.float64, device=DEV)
                call(src, P, 0, None, 0, y, part=part)
                y2 = empty_view_bf(*oshape, ld=cout + 8, off=8)
                call(src, P, 0, None, 0, y2)
                torch.cuda.synchronize()
                assert torch.equal(y.to_nchw(), y2.to_nchw()), (geom, opcode)
                yv = y.to_nchw().double().cpu()
                sums = part.view(N, chunks, cout, 2).sum(1).cpu()
                assert ((sums[..., 0] - yv.sum((2, 3))).abs() <= 1e-5 * yv.abs().sum((2, 3)) + 1e-6).all(), (geom, opcode)
                assert ((sums[..., 1] - (yv * yv).sum((2, 3))).abs() <= 1e-5 * (yv * yv).sum((2, 3)) + 1e-6).all(), (geom, opcode)
            # the activation backward of the layer below in the data-gradient epilogue
            if opcode == 1:
                t = torch.tanh(torch.randn(N, Cb, Hb, Wb, generator=g)).bfloat16().float()       # an activation OUTPUT of the layer below
                vt = to_view_bf(t, ld=Cb + 8, off=8)
                fused = empty_view_bf(*oshape, ld=cout + 8, off=8)
                if op.mul_ok(src, fused, vt):
                    extras['mul'] += 1
                    op.small2big(src, P, 0, None, 0, fused, mul=(vt, L.ACT_TANH))
                    lin = empty_view(*oshape)
                    ref.small2big(src_f, P, 0, None, 0, lin)
                    torch.cuda.synchronize()
                    wantm = lin.to_nchw().double().cpu() * (1 - t.double() ** 2)
                    assert close_bf(fused.to_nchw(), wantm), (geom, 'mul')
    print('window kernel launches checked:', ran, extras)
    assert ran[0] >= 5 and ran[1] >= 5, ran
    assert extras['mul'] >= 3 and extras['part'] >= 3 and extras['split'] >= 2, extras
