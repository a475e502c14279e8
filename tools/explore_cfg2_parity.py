"""Exploration script: at the benchmark configuration (cfg2, bs 16) compare, per step and per loss scalar,
   (a) the fp32 CPU oracle, (b) the HIP path, against (c) the oracle run in float64 on the GPU (torch ops in double).
Usage on the GPU box: python tools/explore_cfg2_parity.py [steps] [batch]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import patchgan_amd as pg
from oracle import patchgan_oracle as O

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
# optional: image size, output classes (> 1: softmax head + weighted BCE, BASELINE config 4), e.g. `... 5 4 512 4`
SIZE = int(sys.argv[3]) if len(sys.argv) > 3 else 256
COUT = int(sys.argv[4]) if len(sys.argv) > 4 else 1
FINAL, LOSS = ('sigmoid', 'tversky') if COUT == 1 else ('softmax', 'weighted_bce')
KEYS = ['gen', 'gen_loss', 'gdisc', 'discr', 'discf', 'disc']
torch.set_num_threads(16)
torch.manual_seed(1234)
g = pg.UNet(3, COUT, 64, use_dropout=False, activation='leakyrelu', final_act=FINAL)
d = pg.Discriminator(3 + COUT, 64, n_layers=3)
gw = {k: v.clone() for k, v in g.state_dict().items()}
dw = {k: v.clone() for k, v in d.state_dict().items()}
gen = torch.Generator().manual_seed(7)
x = torch.rand(B, 3, SIZE, SIZE, generator=gen)
y = (torch.rand(B, COUT, SIZE, SIZE, generator=gen) > 0.7).float()
kw = dict(activation='leakyrelu', final_act=FINAL, n_layers=3, norm=False, loss_type=LOSS)


def run(tr, xx, yy):
    out = []
    for s in range(steps):
        l = tr.batch(xx, yy, train=True)
        out.append([float(l[k]) for k in KEYS])
        print('  step', s + 1, out[-1], flush=True)
    return np.array(out)


print('fp64 oracle on the GPU (torch double)', flush=True)
c64 = run(O.OracleTrainer({k: v.cuda() for k, v in gw.items()}, {k: v.cuda() for k, v in dw.items()}, dtype=torch.float64, **kw),
          x.cuda(), y.cuda())
print('fp32 oracle on the GPU (torch / MIOpen float)', flush=True)
g32 = run(O.OracleTrainer({k: v.cuda() for k, v in gw.items()}, {k: v.cuda() for k, v in dw.items()}, **kw), x.cuda(), y.cuda())
print('HIP path', flush=True)
t = pg.Trainer(g.cuda(), d.cuda(), tempfile.mkdtemp())
t.loss_type = LOSS
t.setup_optimizers(1e-3, 1e-3)
g.train(); d.train()
hip = run(t, x, y)
print('fp32 oracle on the CPU', flush=True)
c32 = run(O.OracleTrainer(gw, dw, **kw), x, y)


def rel(a, b):
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-3)


np.set_printoptions(precision=2, linewidth=200)
for name, c in (('CPU fp32 oracle', c32), ('GPU fp32 torch', g32), ('HIP path', hip)):
    print(name, 'vs fp64: max over scalars per step', rel(c, c64).max(axis=1))
print('HIP vs CPU fp32 per step', rel(hip, c32).max(axis=1))
print('per-scalar HIP vs fp64 at step 1', rel(hip, c64)[0], ' CPU fp32 vs fp64 at step 1', rel(c32, c64)[0])
