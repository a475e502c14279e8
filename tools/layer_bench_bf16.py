"""Per-layer timing of the PG_ALGO_BF16 conv kernels on bf16 tensors at the cfg4 shapes (512x512, nf = ndf = 64, B = 8), the
LDS-DMA kernels of conv_bf16.hip next to the register-staged ones (PG_TUNE_BF16X_OFF), in one process on one device.
    python tools/layer_bench_bf16.py [layer-name-prefix ...]        LB_BATCH / LB_REPS / LB_SIZE override 8 / 10 / 512
Tile / split sweeps: PATCHGAN_EXPERIMENT=1 PATCHGAN_BF16X_TILE=0|1|2 PATCHGAN_BF16X_TARGET=<workgroups> (also _WTILE / _WTARGET)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchgan_amd import engine as E
from patchgan_amd import _lib as L

B = int(os.environ.get('LB_BATCH', '8'))
REPS = int(os.environ.get('LB_REPS', '10'))
S = int(os.environ.get('LB_SIZE', '512'))
dev = torch.device('cuda')
nf = 64
layers = []
h, prev = S // 2, nf
for i, f in enumerate([2 * nf, 4 * nf, 8 * nf, 8 * nf, 8 * nf, 8 * nf], start=1):
    layers.append((f'enc{i}', B, h, h, f, prev, 2))
    prev = f
    h //= 2
dec = [(8 * nf, 8 * nf), (16 * nf, 8 * nf), (16 * nf, 8 * nf), (16 * nf, 4 * nf), (8 * nf, 2 * nf), (4 * nf, nf)]
h = S // 128
for i, (a, b) in enumerate(dec):
    layers.append((f'dec{i}', B, 2 * h, 2 * h, a, b, 2))
    h *= 2
for nb, tag in ((B, 'N'), (2 * B, '2N')):
    layers += [(f'd1/{tag}', nb, S // 2, S // 2, 128, 64, 2), (f'd2/{tag}', nb, S // 4, S // 4, 256, 128, 2),
               (f'd3/{tag}', nb, S // 8, S // 8, 512, 256, 1)]
only = sys.argv[1:]
print(f"{'layer':9s} {'geom':30s} {'GFLOP':>7s} | " + ' | '.join(f"{o:>50s}" for o in ('big2small  flat / ring / regs', 'small2big  flat / ring / regs', 'wgrad')))
tot = [[0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]]
for name, N, Hb, Wb, Ca, Cb, s in layers:
    if only and not any(name.startswith(o) for o in only):
        continue
    ops = [E.ConvOp(N, Hb, Wb, Ca, Cb, s, L.ALGO_BF16 | L.TUNE_BF16X_FLAT), E.ConvOp(N, Hb, Wb, Ca, Cb, s, L.ALGO_BF16 | L.TUNE_BF16X_RING),
           E.ConvOp(N, Hb, Wb, Ca, Cb, s, L.ALGO_BF16 | L.TUNE_BF16X_OFF)]
    big = E.View.alloc(N, Hb, Wb, Cb, dev, bf=True)
    big.t.normal_()
    small = E.View.alloc(N, ops[0].Hs, ops[0].Ws, Ca, dev, bf=True)
    small.t.normal_()
    P = torch.randn(16 * Ca * Cb, device=dev) * 0.05
    dP = torch.empty_like(P)
    io = L.IO_BIG_BF16 | L.IO_SMALL_BF16
    cells = []
    for oc in (0, 1, 2):
        cell = []
        for k, op in enumerate(ops):
            if oc == 2 and k >= 1:
                continue
            fn = {0: lambda: op.big2small(big, P, 0, None, 0, small), 1: lambda: op.small2big(small, P, 0, None, 0, big),
                  2: lambda: op.wgrad(small, big, dP, 0)}[oc]
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / REPS
            sym, split = op.describe(oc, io)
            tot[oc][k] += ms
            tile = sym.split('<')[1][:-1].replace(',true', '').replace(',false', '')[:7]
            cell.append(f"{ms * 1e3:4.0f}us {op.flops / ms / 1e9:4.0f}TF" + (f" {tile}{'/' + str(split) if split > 1 else ''}" if k != 1 else ''))
        cells.append(f"{'  '.join(cell):>50s}")
    print(f"{name:9s} {str((N, Hb, Wb, Ca, Cb, s)):30s} {ops[0].flops / 1e9:7.2f} | " + ' | '.join(cells), flush=True)
print('sum ms: big2small flat %.3f ring %.3f regs %.3f | small2big flat %.3f ring %.3f regs %.3f | wgrad %.3f' % (*tot[0], *tot[1], tot[2][0]))
