"""Per-layer timing of the three conv kernels on the cfg2 shapes (GPU box).  Prints TFLOP/s per layer and op."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchgan_amd import engine as E
from patchgan_amd import _lib as L

B = int(os.environ.get('LB_BATCH', '16'))
REPS = int(os.environ.get('LB_REPS', '10'))
dev = torch.device('cuda')
# name, N, Hb, Wb, Ca, Cb, stride, ops
nf = 64
layers = []
h = 256
prev = 3
for i, f in enumerate([nf, 2 * nf, 4 * nf, 8 * nf, 8 * nf, 8 * nf, 8 * nf]):
    layers.append((f'enc{i}', B, h, h, f, prev, 2, 'b2s(fwd) wgrad' + (' s2b(dgrad)' if i else '')))
    prev = f
    h //= 2
dec = [(8 * nf, 8 * nf), (16 * nf, 8 * nf), (16 * nf, 8 * nf), (16 * nf, 4 * nf), (8 * nf, 2 * nf), (4 * nf, nf), (2 * nf, 1)]
h = 2
for i, (a, b) in enumerate(dec):
    layers.append((f'dec{i}', B, 2 * h, 2 * h, a, b, 2, 's2b(fwd) wgrad b2s(dgrad)'))
    h *= 2
for nb, tag in ((B, 'N'), (2 * B, '2N')):
    layers += [(f'd0/{tag}', nb, 256, 256, 64, 4, 2, 'b2s(fwd) wgrad s2b(dgrad)'), (f'd1/{tag}', nb, 128, 128, 128, 64, 2, 'b2s wgrad s2b'),
               (f'd2/{tag}', nb, 64, 64, 256, 128, 2, 'b2s wgrad s2b'), (f'd3/{tag}', nb, 32, 32, 512, 256, 1, 'b2s wgrad s2b'),
               (f'd4/{tag}', nb, 31, 31, 1, 512, 1, 'b2s wgrad s2b')]
only = sys.argv[1:] 
print(f"{'layer':10s} {'geom':34s} {'GFLOP':>8s} | " + ' | '.join(f"{o:>22s}" for o in ('big2small', 'small2big', 'wgrad')))
tot = {0: 0.0, 1: 0.0, 2: 0.0}
for name, N, Hb, Wb, Ca, Cb, s, _ in layers:
    if only and not any(name.startswith(o) for o in only):
        continue
    op = E.ConvOp(N, Hb, Wb, Ca, Cb, s, E.DEFAULT_ALGO)
    big = E.View.alloc(N, Hb, Wb, Cb, dev); big.t.normal_()
    small = E.View.alloc(N, op.Hs, op.Ws, Ca, dev); small.t.normal_()
    P = torch.randn(16 * Ca * Cb, device=dev) * 0.05
    dP = torch.empty_like(P)
    fns = {0: lambda: op.big2small(big, P, 0, None, 0, small), 1: lambda: op.small2big(small, P, 0, None, 0, big),
           2: lambda: op.wgrad(small, big, dP, 0)}
    cells = []
    for oc in (0, 1, 2):
        fns[oc](); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            fns[oc]()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / REPS
        sym, split = op.describe(oc)
        tot[oc] += ms
        cells.append(f"{ms*1e3:7.0f}us {op.flops/ms/1e9:6.1f}TF {(sym.split('<')[1][:-1] if '<' in sym else sym[-6:])}{'/' + str(split) if split > 1 else '':4s}")
    print(f"{name:10s} {str((N, Hb, Wb, Ca, Cb, s)):34s} {op.flops/1e9:8.2f} | " + ' | '.join(cells), flush=True)
print('sum ms', tot)
