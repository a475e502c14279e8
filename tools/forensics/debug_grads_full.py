"""Debug aid (GPU box): per-parameter step-1 gradient error of the HIP step at full width (nf = ndf = 64, 256x256, B from argv)
against the float64 oracle (torch double ops on the GPU), next to the fp32 CPU oracle's own error; for several tunings."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, tempfile
torch.set_num_threads(16)
from oracle import patchgan_oracle as O
import patchgan_amd as pg
from patchgan_amd import _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
torch.manual_seed(1234)
g0 = pg.UNet(3, 1, 64, use_dropout=False, activation='leakyrelu', final_act='sigmoid')
d0 = pg.Discriminator(4, 64, n_layers=3)
gw = {k: v.clone() for k, v in g0.state_dict().items()}
dw = {k: v.clone() for k, v in d0.state_dict().items()}
gen = torch.Generator().manual_seed(7)
x = torch.rand(B, 3, S, S, generator=gen)
y = (torch.rand(B, 1, S, S, generator=gen) > 0.7).float()
kw = dict(activation='leakyrelu', final_act='sigmoid', n_layers=3, norm=False, loss_type='tversky')
o64 = O.OracleTrainer({k: v.cuda() for k, v in gw.items()}, {k: v.cuda() for k, v in dw.items()}, dtype=torch.float64, **kw)
o64.batch(x.cuda(), y.cuda(), train=True)
o32 = O.OracleTrainer(gw, dw, **kw)
o32.batch(x, y, train=True)
rel = lambda a, b: ((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max()).item()
l2 = lambda a, b: ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm()).item()
res = {}
for name, algo, bits in (('auto', None, 0), ('no_wino', None, L.TUNE_WINO_OFF), ('direct', L.ALGO_DIRECT, 0)):
    g = pg.UNet(3, 1, 64, use_dropout=False, activation='leakyrelu', final_act='sigmoid')
    d = pg.Discriminator(4, 64, n_layers=3)
    g.load_state_dict(gw); d.load_state_dict(dw)
    if algo is not None:
        g.engine.algo = algo; d.engine.algo = algo
    g.set_tuning(bits); d.set_tuning(bits)
    t = pg.Trainer(g.cuda(), d.cuda(), tempfile.mkdtemp()); t.setup_optimizers()
    g.train(); d.train(); t.batch(x, y, train=True)
    res[name] = ({k: p.grad.clone() for k, p in g.named_parameters()}, {k: p.grad.clone() for k, p in d.named_parameters()})
for i, which in enumerate(('g_grads', 'd_grads')):
    for k, w in o64.last[which].items():
        print(f"{which[0]} {k:36s} cpu32 {rel(o32.last[which][k], w):.1e} | " + ' '.join(f"{n} {rel(res[n][i][k], w):.1e}" for n in res) +
              f" | auto-vs-direct {rel(res['auto'][i][k], res['direct'][i][k]):.1e} |g|max {w.abs().max().item():.2e}"
              f" || L2: cpu32 {l2(o32.last[which][k], w):.1e} " + ' '.join(f"{n} {l2(res[n][i][k], w):.1e}" for n in res))
