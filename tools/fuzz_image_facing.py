"""One-off fuzz of the image-facing paths (few channels on the big side: the persistent forward kernel k_b2s_tapkp with and without the
InstanceNorm partial sums, the persistent weight gradient k_wgrad_tapnp, the taps-in-N data gradient + col2im, the one-channel heads
k_s2b_ca1 / k_s2b_ca1_s1) on random geometries -- odd sizes, channel slices of wider buffers, ragged tails -- against the scalar
direct kernels of the same library (PG_ALGO_DIRECT).  tools/fuzz_image_facing.py [cases]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchgan_amd import engine as E, _lib as L

dev = torch.device('cuda')
random.seed(int(os.environ.get('SEED', '0')))
torch.manual_seed(0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
worst = [0.0, 0.0, 0.0, 0.0]
paths = {}


def sliced(N, H, W, C):
    """A view of C channels inside a wider pixel (as x | mask live in the discriminator-input buffer) or a dense one."""
    if random.random() < 0.5:
        return E.View.alloc(N, H, W, C, dev)
    ld = C + random.choice([1, 2, 4, 5])
    wide = E.View.alloc(N, H, W, ld, dev)
    wide.t.normal_()
    return wide.channels(random.randint(0, ld - C), C)


for it in range(cases):
    kind = random.choice(['s2', 's2', 's2', 'head'])
    N = random.choice([1, 2, 3, 5, 8, 16, 32])
    if kind == 's2':
        stride = 2
        Hb, Wb = random.choice([16, 30, 33, 64, 100, 128, 129, 256]), random.choice([16, 18, 31, 64, 96, 128, 200, 256])
        Cb = random.choice([1, 2, 3, 4, 4, 5, 7, 8])
        Ca = random.choice([4, 8, 12, 16, 32, 64, 64, 128, 6, 10])
    else:                                                # the discriminator's one-channel head: stride 1 onto Ca = 1
        stride = 1
        Hb, Wb = random.randint(5, 40), random.randint(5, 40)
        Cb = random.choice([8, 64, 128, 512])
        Ca = 1
    if N * Hb * Wb * max(Ca, Cb) > (1 << 27):
        N = max(1, N // 8)
    auto, ref = E.ConvOp(N, Hb, Wb, Ca, Cb, stride, L.ALGO_AUTO), E.ConvOp(N, Hb, Wb, Ca, Cb, stride, L.ALGO_DIRECT)
    Hs, Ws = auto.Hs, auto.Ws
    big = sliced(N, Hb, Wb, Cb) if kind == 's2' else E.View.alloc(N, Hb, Wb, Cb, dev)
    big.from_nchw(torch.randn(N, Cb, Hb, Wb, device=dev))
    small = E.View.alloc(N, Hs, Ws, Ca, dev)
    small.t.normal_()
    P = torch.randn(16 * Ca * Cb, device=dev) * (1.0 / (16 * Cb) ** 0.5)
    bias_a, bias_b = torch.randn(Ca, device=dev), torch.randn(Cb, device=dev)
    act = random.choice([L.ACT_NONE, L.ACT_CODES['leakyrelu'], L.ACT_CODES['tanh']])
    outs = []
    stats = None
    for op in (auto, ref):
        o0 = E.View.alloc(N, Hs, Ws, Ca, dev)
        use_bias = random.random() < 0.5 if op is auto else use_bias
        kw = {}
        if op is auto and act in (L.ACT_NONE,) and not use_bias:
            nch = auto.stats_chunks(0, big, o0)
            if nch:
                part = torch.full((N, nch, Ca, 2), float('nan'), dtype=torch.float64, device=dev)
                kw['part'] = part
        op.big2small(big, P, 0, bias_a if use_bias else None, 0, o0, act, **kw)
        if kw:
            stats = (kw['part'].sum(1), o0.to_nchw().double())
        o1 = sliced(N, Hb, Wb, Cb) if (op is auto and kind == 's2') else E.View.alloc(N, Hb, Wb, Cb, dev)
        op.small2big(small, P, 0, bias_b if use_bias else None, 0, o1)
        dP = torch.empty_like(P)
        db = torch.empty(Ca, device=dev)
        op.wgrad(small, big, dP, 0, dbias=db)
        outs.append((o0.to_nchw().clone(), o1.to_nchw().clone(), dP.clone(), db.clone()))
    torch.cuda.synchronize()
    errs = [((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item() for a, b in zip(*outs)]
    key = tuple(auto.describe(oc)[0] for oc in (0, 1, 2)) + (stats is not None,)
    paths[key] = paths.get(key, 0) + 1
    worst = [max(w, e) for w, e in zip(worst, errs)]
    geom = (kind, N, Hb, Wb, Ca, Cb, big.ld, act)
    assert errs[0] < 2e-5 and errs[1] < 2e-5 and errs[2] < 5e-5 and errs[3] < 5e-5, (geom, errs, key)
    if stats is not None:
        s, o = stats
        want = torch.stack((o.sum((2, 3)), (o * o).sum((2, 3))), -1)
        e = ((s - want).abs().max() / want.abs().max()).item()
        assert e < 2e-6, (geom, 'stats', e)
print('cases per path:')
for k, v in sorted(paths.items(), key=lambda kv: -kv[1]):
    print(' ', v, k)
print('worst relative max-norm error (forward, data gradient, weight gradient, bias gradient):', worst)
