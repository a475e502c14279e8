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
