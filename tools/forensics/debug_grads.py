"""Debug aid (GPU box): per-parameter gradient error of the HIP step vs the fp64 / fp32 CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, tempfile
torch.set_num_threads(16)
from oracle import patchgan_oracle as O
from tests.golden_util import Golden
import patchgan_amd as pg
name = sys.argv[1] if len(sys.argv) > 1 else 'b_tanh_wbce_norm'
gold = Golden(name); c = gold.cfg
x, y = gold.inputs()
def orc(dtype):
    ot = O.OracleTrainer(gold.weights('g0'), gold.weights('d0'), activation=c['activation'], final_act=c['final_act'], n_layers=c['n_layers'], norm=c['norm'], loss_type=c['loss_type'], dtype=dtype)
    ot.batch(x, y, train=True); return ot.last
o32, o64 = orc(torch.float32), orc(torch.float64)
g = pg.UNet(c['in_nc'], c['out_nc'], c['nf'], activation=c['activation'], final_act=c['final_act'])
d = pg.Discriminator(c['in_nc'] + c['out_nc'], c['ndf'], n_layers=c['n_layers'], norm=c['norm'])
g.load_state_dict(gold.weights('g0')); d.load_state_dict(gold.weights('d0')); g.cuda(); d.cuda()
t = pg.Trainer(g, d, tempfile.mkdtemp()); t.loss_type = c['loss_type']; t.setup_optimizers()
g.train(); d.train(); t.batch(x, y, train=True)
rel = lambda a, b: ((a.double().cpu() - b.double()).abs().max() / b.double().abs().max()).item()
for which, mod in (('g_grads', g), ('d_grads', d)):
    for k, w in o64[which].items():
        print(f"{which} {k:40s} hip-vs-64 {rel(mod.get_parameter(k).grad, w):.3e}  cpu32-vs-64 {rel(o32[which][k], w):.3e}  |g|max {w.abs().max().item():.3e}")
