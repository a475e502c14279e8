"""One-off fuzz of the stride-1 Winograd paths (F(3x3,4x4) row-split / fused / K-split, F(2x2,4x4), F(4x4,3x3) / F(4x4,2x2) weight gradient,
the V hand-over) on random geometries against the exact implicit GEMM of the same library (PG_ALGO_MFMA).  tools/fuzz_stride1.py [cases]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchgan_amd import engine as E, _lib as L

dev = torch.device('cuda')
random.seed(int(os.environ.get('SEED', '0')))
torch.manual_seed(0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
worst = [0.0, 0.0, 0.0]
paths = {}
for it in range(cases):
    N = random.choice([4, 6, 9, 12, 16, 24])
    Hb, Wb = random.randint(24, 72), random.randint(24, 72)
    Ca, Cb = random.choice([64, 96, 128, 160, 256, 512]), random.choice([64, 128, 192, 256])
    bits = random.choice([0, 0, L.TUNE_WINO1_F2, L.TUNE_WINO1_F3])
    auto, ref = E.ConvOp(N, Hb, Wb, Ca, Cb, 1, L.ALGO_AUTO | bits), E.ConvOp(N, Hb, Wb, Ca, Cb, 1, L.ALGO_MFMA)
    if not auto.describe(0)[0].startswith('k_wino'):
        continue
    Hs, Ws = auto.Hs, auto.Ws
    big = E.View.alloc(N, Hb, Wb, Cb, dev); big.t.normal_()
    small = E.View.alloc(N, Hs, Ws, Ca, dev); small.t.normal_()
    P = torch.randn(16 * Ca * Cb, device=dev) * (1.0 / (16 * Cb) ** 0.5)
    bias_a, bias_b = torch.randn(Ca, device=dev), torch.randn(Cb, device=dev)
    outs = []
    for op in (auto, ref):
        o0 = E.View.alloc(N, Hs, Ws, Ca, dev)
        vb = auto.v_bytes() if op is auto else 0
        vk = torch.empty(vb, dtype=torch.uint8, device=dev) if vb else None
        op.big2small(big, P, 0, bias_a, 0, o0, L.ACT_CODES['leakyrelu'], **({'v_keep': vk} if vk is not None else {}))
        o1 = E.View.alloc(N, Hb, Wb, Cb, dev)
        op.small2big(small, P, 0, bias_b, 0, o1)
        dP = torch.empty_like(P)
        op.wgrad(small, big, dP, 0, **({'v_pre': vk} if vk is not None else {}))
        outs.append((o0.t.clone(), o1.t.clone(), dP.clone()))
    torch.cuda.synchronize()
    errs = [((a - b).abs().max() / b.abs().max()).item() for a, b in zip(*outs)]
    key = tuple(auto.describe(oc)[0] for oc in (0, 1, 2)) + (bool(auto.v_bytes()),)
    paths[key] = paths.get(key, 0) + 1
    worst = [max(w, e) for w, e in zip(worst, errs)]
    assert errs[0] < 2e-5 and errs[1] < 2e-5 and errs[2] < 5e-5, ((N, Hb, Wb, Ca, Cb), bits, errs, key)
print('cases per path:')
for k, v in sorted(paths.items(), key=lambda kv: -kv[1]):
    print(' ', v, k)
print('worst relative max-norm error (forward, data gradient, weight gradient):', worst)
