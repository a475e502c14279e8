"""Which tensor of a training step differs from run to run while another process loads the GPU?  The network of tools/debug_repro.py,
one step per repetition from the same state; the per-layer outputs of every discriminator forward, the generator output and the
gradient buffers are recorded and compared bitwise with the first repetition.
usage: python tools/debug_race_step.py [bf16|fp32] [N] [REPS]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import patchgan_amd as pg
from patchgan_amd import engine as E

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
torch.manual_seed(77)
g0 = pg.UNet(3, 4, 64, activation='leakyrelu', final_act='softmax', use_dropout=False)
d0 = pg.Discriminator(7, 64, n_layers=3)
gw = {k: v.clone() for k, v in g0.state_dict().items()}
dw = {k: v.clone() for k, v in d0.state_dict().items()}
gen = torch.Generator().manual_seed(8)
x = torch.rand(N, 3, 256, 256, generator=gen)
y = (torch.rand(N, 4, 256, 256, generator=gen) > 0.7).float()

rec = []
orig_df, orig_db, orig_gf, orig_gb = E.DiscriminatorEngine.forward, E.DiscriminatorEngine.backward, E.GeneratorEngine.forward, E.GeneratorEngine.backward


def df(self, flat, din, *a, **k):
    c = orig_df(self, flat, din, *a, **k)
    tag = f'D.forward#{sum(1 for n, _ in rec if n.startswith("D.forward") and n.endswith("/in"))}'
    rec.append((tag + '/in', din.t.clone()))
    for i, (s, t) in enumerate(zip(c.src, c.t)):
        rec.append((f'{tag}/src{i}', s.t.clone()))
        rec.append((f'{tag}/t{i}', t.t.clone()))
    return c


def db(self, flat, gflat, c, gout, *a, **k):
    r = orig_db(self, flat, gflat, c, gout, *a, **k)
    tag = f'D.backward#{sum(1 for n, _ in rec if n.startswith("D.backward") and n.endswith("/gout"))}'
    rec.append((tag + '/gout', gout.t.clone()))
    if r is not None:
        rec.append((tag + '/dx', r.t.clone()))
    if gflat is not None:
        rec.append((tag + '/gflat', gflat.clone()))
    return r


def gf(self, flat, xin, gen_out, *a, **k):
    c = orig_gf(self, flat, xin, gen_out, *a, **k)
    rec.append(('G.forward/out', gen_out.t.clone()))
    return c


def gb(self, flat, gflat, *a, **k):
    r = orig_gb(self, flat, gflat, *a, **k)
    rec.append(('G.backward/gflat', gflat.clone()))
    return r


E.DiscriminatorEngine.forward, E.DiscriminatorEngine.backward, E.GeneratorEngine.forward, E.GeneratorEngine.backward = df, db, gf, gb
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
    rec.clear()
    l = t.batch(x, y, train=True)
    t.flush()
    torch.cuda.synchronize()
    rec.append(('losses', torch.tensor([l[k] for k in ('gen', 'gen_loss', 'gdisc', 'discr', 'discf', 'disc')], dtype=torch.float64)))
    rec.append(('G.flat after', g.flat.clone()))
    rec.append(('D.flat after', d.flat.clone()))
    runs.append(list(rec))
names = [n for n, _ in runs[0]]
for r in range(1, reps):
    first = None
    for (n0, a), (n1, b) in zip(runs[0], runs[r]):
        assert n0 == n1
        same = a.shape == b.shape and torch.equal(a.contiguous().view(torch.uint8), b.contiguous().view(torch.uint8))
        if not same:
            af, bf = a.double(), b.double()
            m = ~(torch.isnan(af) | torch.isnan(bf))
            rel = float((af[m] - bf[m]).abs().max() / af[m].abs().max().clamp_min(1e-30)) if m.any() else float('nan')
            nd = int(((af != bf) & m).sum())
            first = (n0, rel, nd, a.numel())
            break
    print(f'rep {r} vs 0:', 'all equal' if first is None else f'FIRST DIFFERENCE at {first[0]}: rel {first[1]:.2e}, {first[2]} of {first[3]} elements', flush=True)
