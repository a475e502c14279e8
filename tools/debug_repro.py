"""Is a training step bitwise reproducible run to run?  The network of tests/test_dp_gpu.py::test_two_rank_bf16_storage_tracks_single_process
(4-class softmax head, weighted BCE, nf = ndf = 64) for `precision` in fp32 / bf16, batch N, 3 steps, REPS times from the same state.
usage: python tools/debug_repro.py [bf16|fp32] [N] [REPS]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import patchgan_amd as pg

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
KEYS = ['gen', 'gen_loss', 'gdisc', 'discr', 'discf', 'disc']
torch.manual_seed(77)
g0 = pg.UNet(3, 4, 64, activation='leakyrelu', final_act='softmax', use_dropout=False)
d0 = pg.Discriminator(7, 64, n_layers=3)
gw = {k: v.clone() for k, v in g0.state_dict().items()}
dw = {k: v.clone() for k, v in d0.state_dict().items()}
gen = torch.Generator().manual_seed(8)
x = torch.rand(N, 3, 256, 256, generator=gen)
y = (torch.rand(N, 4, 256, 256, generator=gen) > 0.7).float()
runs = []
for r in range(reps):
    g = pg.UNet(3, 4, 64, activation='leakyrelu', final_act='softmax', use_dropout=False)
    d = pg.Discriminator(7, 64, n_layers=3)
    g.load_state_dict(gw); d.load_state_dict(dw)
    g.cuda().set_precision(prec); d.cuda().set_precision(prec)
    t = pg.Trainer(g, d, tempfile.mkdtemp())
    t.loss_type = 'weighted_bce'
    t.setup_optimizers(1e-3, 1e-3)
    g.train(); d.train()
    ls = np.array([[t.batch(x, y, train=True)[k] for k in KEYS] for _ in range(3)])      # (one Trainer.batch call per KEY: 18 steps, rows of six)
    torch.cuda.synchronize()
    runs.append((ls, g.flat.clone(), d.flat.clone()))
    print(f'run {r}: step-1 losses {ls[0]}', flush=True)
for r in range(1, reps):
    same = np.array_equal(runs[0][0], runs[r][0]) and torch.equal(runs[0][1], runs[r][1]) and torch.equal(runs[0][2], runs[r][2])
    rel = np.abs(runs[r][0] - runs[0][0]) / np.maximum(np.abs(runs[0][0]), 1e-3)
    print(f'run {r} vs run 0: bitwise {"EQUAL" if same else "DIFFERENT"}; max rel loss difference per step {rel.max(axis=1)}')
