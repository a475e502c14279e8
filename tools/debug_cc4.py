"""As tools/debug_cc3.py, but only the discriminator head's data-gradient call (Ca == 1) is recorded: its input dy, its multiplier t and
its output, cloned on the device.  With PATCHGAN_EXPERIMENT=1 PATCHGAN_CA1S1_BF16=1 the LDS-staged kernel runs on bf16 outputs.
Two of these at once.  usage: python tools/debug_cc4.py [reps] [steps]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import patchgan_amd as pg
from patchgan_amd import engine as E, _lib as L

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 18
N = 1
rec = []
o_s2b = E.ConvOp.small2big


def s2b(self, small, P, p_off, bias, b_off, big, act=L.ACT_NONE, **kw):
    if self.Ca != 1:
        return o_s2b(self, small, P, p_off, bias, b_off, big, act, **kw)
    i = len(rec) // 4
    rec.append((f'call {i} input dy', small.t.clone()))
    mul = kw.get('mul')
    rec.append((f'call {i} multiplier t', mul[0].t.clone() if mul else torch.zeros(1, device='cuda')))
    rec.append((f'call {i} weights', P[p_off:p_off + 16 * self.Cb].clone()))
    r = o_s2b(self, small, P, p_off, bias, b_off, big, act, **kw)
    rec.append((f'call {i} OUTPUT {tuple(big.t.shape)} {big.t.dtype}', big.t.clone()))
    return r


E.ConvOp.small2big = s2b
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
    g.cuda().set_precision('bf16'); d.cuda().set_precision('bf16')
    t = pg.Trainer(g, d, tempfile.mkdtemp())
    t.loss_type = 'weighted_bce'
    t.setup_optimizers(1e-3, 1e-3)
    g.train(); d.train()
    rec = []
    for s in range(steps):
        l = t.batch(x, y, train=True)
        l['gen']
    runs.append(rec)
torch.cuda.synchronize()
for r in range(1, reps):
    first = None
    for (k, a), (_, b) in zip(runs[0], runs[r]):
        if a.shape != b.shape or not torch.equal(a.contiguous().view(torch.uint8), b.contiguous().view(torch.uint8)):
            af, bf = a.double().flatten(), b.double().flatten()
            idx = torch.nonzero(af != bf).flatten()
            first = f'{k}: {idx.numel()} of {af.numel()} elements differ; first indices {idx[:12].tolist()} last {idx[-3:].tolist()}; max abs diff {float((af - bf).abs().max()):.3e} (max |value| {float(af.abs().max()):.3e})'
            break
    print(f'pid {os.getpid()} rep {r}:', 'all equal' if first is None else 'FIRST: ' + first, flush=True)
