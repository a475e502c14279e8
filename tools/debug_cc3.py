"""As tools/debug_cc2.py, but every conv call's OUTPUT is cloned on the device right after the call (no synchronisation): the first
call whose output differs from repetition 0 names the kernel.  Two of these at once on one GPU.
usage: python tools/debug_cc3.py [precision] [reps] [steps]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import patchgan_amd as pg
from patchgan_amd import engine as E, _lib as L

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
N = 1
rec = []
o_b2s, o_s2b, o_wg, o_bb = E.ConvOp.big2small, E.ConvOp.small2big, E.ConvOp.wgrad, E.ConvOp.bwd_big


def geom(op):
    return (op.N, op.Hb, op.Wb, op.Ca, op.Cb, op.stride)


def b2s(self, big, P, p_off, bias, b_off, small, act=L.ACT_NONE, **kw):
    r = o_b2s(self, big, P, p_off, bias, b_off, small, act, **kw)
    rec.append((f'#{len(rec)} b2s {geom(self)} {self.describe(0, self._io(big, small))[0]} kw={sorted(k for k, v in kw.items() if v is not None)}', small.t.clone()))
    if kw.get('part') is not None:
        rec.append((f'#{len(rec)}   its stats partials', kw['part'].clone()))
    return r


def s2b(self, small, P, p_off, bias, b_off, big, act=L.ACT_NONE, **kw):
    r = o_s2b(self, small, P, p_off, bias, b_off, big, act, **kw)
    rec.append((f'#{len(rec)} s2b {geom(self)} {self.describe(1, self._io(big, small))[0]} kw={sorted(k for k, v in kw.items() if v is not None)}', big.t.clone()))
    if kw.get('part') is not None:
        rec.append((f'#{len(rec)}   its stats partials', kw['part'].clone()))
    return r


def wg(self, small, big, dP, p_off, dbias=None, b_off=0, **kw):
    r = o_wg(self, small, big, dP, p_off, dbias, b_off, **kw)
    n = 16 * self.Ca * self.Cb
    rec.append((f'#{len(rec)} wgrad {geom(self)} {self.describe(2, self._io(big, small))[0]}', dP[p_off:p_off + n].clone()))
    return r


def bb(self, small, big, P, dP, p_off, dsmall, **kw):
    r = o_bb(self, small, big, P, dP, p_off, dsmall, **kw)
    n = 16 * self.Ca * self.Cb
    rec.append((f'#{len(rec)} bwd_big dW {geom(self)}', dP[p_off:p_off + n].clone()))
    rec.append((f'#{len(rec)} bwd_big dx {geom(self)}', dsmall.t.clone()))
    return r


E.ConvOp.big2small, E.ConvOp.small2big, E.ConvOp.wgrad, E.ConvOp.bwd_big = b2s, s2b, wg, bb
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
    rec = []
    for s in range(steps):
        rec.append((f'#{len(rec)} ---- step {s + 1} starts: G weights', g.flat.clone()))
        rec.append((f'#{len(rec)} ---- step {s + 1} starts: D weights', d.flat.clone()))
        l = t.batch(x, y, train=True)
        l['gen']
    runs.append(rec)
    if r > 0:       # keep memory bounded: compare now, keep only repetition 0
        torch.cuda.synchronize()
        first, nd = None, 0
        for (k, a), (k2, b) in zip(runs[0], runs[r]):
            if a.shape != b.shape or not torch.equal(a.contiguous().view(torch.uint8), b.contiguous().view(torch.uint8)):
                nd += 1
                if first is None:
                    af, bf = a.double(), b.double()
                    m = ~(torch.isnan(af) | torch.isnan(bf))
                    first = f'{k}: rel {float((af[m] - bf[m]).abs().max() / af[m].abs().max().clamp_min(1e-30)):.1e} ({int(((af != bf) & m).sum())} of {a.numel()} elements)'
        print(f'pid {os.getpid()} rep {r}:', 'all equal' if first is None else f'FIRST: {first}   (+{nd - 1} more)', flush=True)
        runs[r] = None
