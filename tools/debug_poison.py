"""Does a training step read memory it never wrote?  The network of tools/debug_repro.py; before each run the caching allocator's free
blocks are filled with a different pattern (zeros / huge values / NaN), so stale data differs between runs: bitwise-equal losses and
weights across the patterns mean no result depends on uninitialised memory.  PATCHGAN_DP_FORCE=1 adds the data-parallel code path
(one-rank group).  usage: python tools/debug_poison.py [bf16|fp32] [N]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import patchgan_amd as pg

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2
KEYS = ['gen', 'gen_loss', 'gdisc', 'discr', 'discf', 'disc']
if os.environ.get('PATCHGAN_DP_FORCE') == '1':
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
    torch.cuda.set_device(0)
    dist.init_process_group(os.environ.get('PATCHGAN_DIST_BACKEND', 'nccl'), rank=0, world_size=1)
torch.manual_seed(77)
g0 = pg.UNet(3, 4, 64, activation='leakyrelu', final_act='softmax', use_dropout=False)
d0 = pg.Discriminator(7, 64, n_layers=3)
gw = {k: v.clone() for k, v in g0.state_dict().items()}
dw = {k: v.clone() for k, v in d0.state_dict().items()}
gen = torch.Generator().manual_seed(8)
x = torch.rand(N, 3, 256, 256, generator=gen)
y = (torch.rand(N, 4, 256, 256, generator=gen) > 0.7).float()
runs = []
for pattern in (0.0, 3.0e38, float('nan'), -1.0):
    junk = [torch.full((1 << 28,), pattern, device='cuda') for _ in range(6)]     # 6 GiB of `pattern`, then back to the allocator's cache
    # ... and blocks of every size class down to 512 bytes (the small pool's 2-MiB segments are separate from the large blocks above)
    for k in range(7, 25):
        junk += [torch.full((1 << k,), pattern, device='cuda') for _ in range(48 if k < 22 else 8)]
    torch.cuda.synchronize()
    del junk
    g = pg.UNet(3, 4, 64, activation='leakyrelu', final_act='softmax', use_dropout=False)
    d = pg.Discriminator(7, 64, n_layers=3)
    g.load_state_dict(gw); d.load_state_dict(dw)
    g.cuda().set_precision(prec); d.cuda().set_precision(prec)
    t = pg.Trainer(g, d, tempfile.mkdtemp())
    t.loss_type = 'weighted_bce'
    t.setup_optimizers(1e-3, 1e-3)
    g.train(); d.train()
    ls = np.array([[t.batch(x, y, train=True)[k] for k in KEYS] for _ in range(3)])      # (one Trainer.batch call per KEY: 18 steps, rows of six)
    t.flush()
    torch.cuda.synchronize()
    runs.append((ls, g.flat.clone(), d.flat.clone()))
    print(f'pattern {pattern}: step-1 losses {ls[0]}', flush=True)
    del g, d, t
for r in range(1, len(runs)):
    same = np.array_equal(runs[0][0], runs[r][0], equal_nan=True) and torch.equal(runs[0][1], runs[r][1]) and torch.equal(runs[0][2], runs[r][2])
    rel = np.abs(runs[r][0] - runs[0][0]) / np.maximum(np.abs(runs[0][0]), 1e-3)
    print(f'run {r} vs run 0: bitwise {"EQUAL" if same else "DIFFERENT"}; max rel loss difference per step {rel.max(axis=1)}')
