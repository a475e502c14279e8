"""Which conv call is not bit-reproducible while ANOTHER process keeps the GPU busy?  Every conv call of bench.py's configurations
(tests/bench_layers.py) is run REPS times on the same operands and the outputs compared bitwise with the first run.
usage: python tools/debug_race.py <cfg> <fp32|bf16> [REPS] [name filter ...]   (start a load first, e.g. tools/soak.py in the background)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import bench_layers as BL
from tests import test_bench_layers_gpu as TB
from patchgan_amd import _lib as L

cfg, mode = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
only = sys.argv[4:]
if os.environ.get('RACE_BATCH'):           # e.g. RACE_BATCH=1: the per-rank geometries of the two-rank tests
    BL.CONFIGS[cfg] = dict(BL.CONFIGS[cfg], batch=int(os.environ['RACE_BATCH']))
    if os.environ.get('RACE_SIZE'):
        BL.CONFIGS[cfg]['size'] = int(os.environ['RACE_SIZE'])
bad = 0
for cs in BL.cases(cfg, mode):
    if only and not any(o in cs.key for o in only):
        continue
    N, Hb, Wb, Ca, Cb, s = cs.geom
    op = cs.convop()
    T = TB._get_truth(cs.geom, mode == 'bf16')
    Hs, Ws = T.Hs, T.Ws
    P = T.W.permute(2, 3, 0, 1).contiguous().reshape(-1).float()
    outs = []
    for r in range(reps + 1):
        if cs.op == 'b2s':
            src = TB._store(T.big, cs.big)
            out = TB._empty(N, Hs, Ws, Ca, cs.small)
            op.big2small(src, P, 0, T.bias_a.float() if cs.bias else None, 0, out, L.ACT_CODES[cs.act])
            res = [out.t.clone()]
            from patchgan_amd import engine as E
            nb = op.u_bytes(0, cs.io) if E.ConvOp.fits(src, out) and (cs.io or E.ConvOp._aligned(src, out)) else 0
            if nb:
                u = torch.zeros(nb, dtype=torch.uint8, device='cuda')
                for valid in (False, True):
                    o3 = TB._empty(N, Hs, Ws, Ca, cs.small)
                    op.big2small(src, P, 0, T.bias_a.float() if cs.bias else None, 0, o3, L.ACT_CODES[cs.act], u_cache=u, u_valid=valid)
                    res.append(o3.t.clone())
            if not cs.bias and cs.act == 'none':
                chunks = op.stats_chunks(0, src, out)
                if chunks:
                    part = torch.full((N * chunks * Ca * 2,), float('nan'), dtype=torch.float64, device='cuda')
                    o2 = TB._empty(N, Hs, Ws, Ca, cs.small)
                    op.big2small(src, P, 0, None, 0, o2, part=part)
                    res += [o2.t.clone(), part.clone()]
            res = tuple(res)
        elif cs.op == 's2b':
            src = TB._store(T.small, cs.small)
            out = TB._empty(N, Hb, Wb, Cb, cs.big)
            op.small2big(src, P, 0, T.bias_b.float() if cs.bias else None, 0, out, L.ACT_CODES[cs.act])
            res = [out.t.clone()]
            # the hand-overs the engines use on this call: the activation backward of the layer below in the epilogue, the packed-weight cache
            if cs.role == 'dgrad' and cs.layer.startswith('d') and not cs.layer.startswith('dec') and not cs.layer.startswith('d0'):
                below = 'leakyrelu' if cs.layer.startswith('d1') else 'tanh'
                t64 = T.t_big if below == 'tanh' else TB._act64(T.t_big * 3 - 1, 'leakyrelu')
                if mode == 'bf16':
                    t64 = t64.float().bfloat16().double()
                tv = TB._store(t64, cs.big)
                o4 = TB._empty(N, Hb, Wb, Cb, cs.big)
                if op.mul_ok(src, o4, tv):
                    op.small2big(src, P, 0, None, 0, o4, mul=(tv, L.ACT_CODES[below]))
                    res.append(o4.t.clone())
            from patchgan_amd import engine as E
            nb = op.u_bytes(1, cs.io) if E.ConvOp.fits(src, out) and (cs.io or E.ConvOp._aligned(src, out)) else 0
            if nb:
                u = torch.zeros(nb, dtype=torch.uint8, device='cuda')
                for valid in (False, True):
                    o3 = TB._empty(N, Hb, Wb, Cb, cs.big)
                    op.small2big(src, P, 0, None, 0, o3, L.ACT_CODES[cs.act], u_cache=u, u_valid=valid)
                    res.append(o3.t.clone())
            res = tuple(res)
        elif cs.op == 'wgrad':
            vs, vb = TB._store(T.small, cs.small), TB._store(T.big, cs.big)
            dP = torch.full((16 * Ca * Cb,), float('nan'), device='cuda')
            op.wgrad(vs, vb, dP, 0, None, 0)
            res = (dP.clone(),)
        else:
            vs, vb = TB._store(T.small, cs.small), TB._store(T.big, cs.big)
            dP = torch.full((16 * Ca * Cb,), float('nan'), device='cuda')
            ds = TB._empty(N, Hs, Ws, Ca, BL.Operand(cs.small.bf))
            op.bwd_big(vs, vb, P, dP, 0, ds)
            res = (dP.clone(), ds.t.clone())
        torch.cuda.synchronize()
        outs.append(res)
    diff = 0
    for r in range(1, reps + 1):
        for a, b in zip(outs[0], outs[r]):
            same = torch.equal(a.view(torch.uint8), b.view(torch.uint8))
            diff += 0 if same else 1
    nan_out = any(bool(torch.isnan(o.float()[:o.numel()]).any()) for o in outs[0][:1]) if cs.op in ('wgrad',) else False
    if diff:
        bad += 1
        a, b = outs[0][-1].double(), None
        worst = 0.0
        for r in range(1, reps + 1):
            for x, y in zip(outs[0], outs[r]):
                d = (x.double() - y.double())
                d = d[~torch.isnan(d)]
                if d.numel():
                    worst = max(worst, float(d.abs().max() / x.double()[~torch.isnan(x.double())].abs().max()))
        print(f'NOT REPRODUCIBLE  {cs.key:40s} {cs.symbols()}  {diff} of {reps} runs differ, worst rel {worst:.2e}', flush=True)
print(f'{cfg} {mode}: {bad} call(s) not reproducible')
