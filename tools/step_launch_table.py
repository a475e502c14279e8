"""Every conv launch of ONE training step in its place in the step (cold operands, neighbours as they are), with the layer geometry:
kernel symbol, split, microseconds (mean of the sampled steps, HIP events on the launch stream), executed and useful TFLOP/s.  The
back-to-back layer benchmarks rank kernels differently; this is the table to choose by.
    python tools/step_launch_table.py [cfg4] [bf16|f32] [steps=6]        PG_TUNE=<bits> sets PG_TUNE_* bits on both networks"""
import os
import sys
import tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import patchgan_amd as pg
from patchgan_amd import engine as E

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'cfg4']
dtype = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device('cuda')
torch.manual_seed(1234)
G = pg.UNet(3, cfg['out_nc'], cfg['nf'], use_dropout=False, activation=cfg['activation'], final_act=cfg['final_act']).to(dev)
D = pg.Discriminator(3 + cfg['out_nc'], cfg['ndf'], n_layers=cfg['n_layers'], norm=False).to(dev)
if dtype == 'bf16':
    G.set_precision('bf16')
    D.set_precision('bf16')
if os.environ.get('PG_TUNE'):            # e.g. PG_TUNE=0x4000 = TUNE_BF16X_RING on every layer
    G.set_tuning(int(os.environ['PG_TUNE'], 0))
    D.set_tuning(int(os.environ['PG_TUNE'], 0))
t = pg.Trainer(G, D, tempfile.mkdtemp(prefix='pgtab_'))
t.loss_type, t.seg_alpha = cfg['loss_type'], 200
t.setup_optimizers(1e-3, 1e-3)
G.train()
D.train()
x, y = bench.make_inputs(cfg['batch'], 0, cfg)
x, y = x.to(dev), y.to(dev)
for _ in range(4):
    t.batch(x, y, train=True)
torch.cuda.synchronize()


class Prof(E.LaunchProfiler):
    def __init__(self):
        super().__init__()
        self.geoms = []

    def launch(self, op, opcode, fn, io=0):
        n = len(self.records)
        r = super().launch(op, opcode, fn, io)
        if len(self.records) > n:
            self.geoms.append((opcode, op.N, op.Hb, op.Wb, op.Ca, op.Cb, op.stride))
        return r

    def launch2(self, op, opcodes, fn, io=0):
        n = len(self.records)
        r = super().launch2(op, opcodes, fn, io)
        for k in range(len(self.records) - n):
            self.geoms.append((opcodes[k] if len(self.records) - n == len(opcodes) else -1, op.N, op.Hb, op.Wb, op.Ca, op.Cb, op.stride))
        return r


runs = []
for _ in range(steps):
    p = Prof()
    E.PROFILER = p
    t.batch(x, y, train=True)
    E.PROFILER = None
    torch.cuda.synchronize()
    runs.append(p)
n = len(runs[0].records)
assert all(len(r.records) == n for r in runs)
tot = ktot = utot = 0.0
print(f"{'#':>3s} {'op':>2s} {'N':>3s} {'Hb':>4s} {'Ca':>5s} {'Cb':>5s} s {'split':>5s} {'us':>8s} {'exec TF/s':>9s} {'useful':>7s}  kernel")
for i in range(n):
    sym, split, flops, kf, uf, _, _ = runs[0].records[i]
    us = sorted(1e3 * r.records[i][5].elapsed_time(r.records[i][6]) for r in runs)[len(runs) // 2]
    oc, N, Hb, Wb, Ca, Cb, s = runs[0].geoms[i]
    tot += us
    ktot += kf
    utot += uf
    print(f'{i:3d} {oc:2d} {N:3d} {Hb:4d} {Ca:5d} {Cb:5d} {s} {split:5d} {us:8.1f} {kf / us / 1e6:9.1f} {uf / us / 1e6:7.1f}  {sym}')
print(f'sum {tot / 1e3:.3f} ms per step in {n} main conv kernels; executed {ktot / tot / 1e6:.1f} TFLOP/s, useful {utot / tot / 1e6:.1f}')
