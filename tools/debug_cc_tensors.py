"""End-of-step tensors of ONE bf16 training step (the network of tools/debug_repro.py, N = 1) against a baseline file: the generator
output, both gradient buffers, both weight buffers -- no kernels added between the step's own launches.
usage: python tools/debug_cc_tensors.py save|check <file> [precision] [reps]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import patchgan_amd as pg

what, path = sys.argv[1], sys.argv[2]
prec = sys.argv[3] if len(sys.argv) > 3 else 'bf16'
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
N = 1
torch.manual_seed(77)
g0 = pg.UNet(3, 4, 64, activation='leakyrelu', final_act='softmax', use_dropout=False)
d0 = pg.Discriminator(7, 64, n_layers=3)
gw = {k: v.clone() for k, v in g0.state_dict().items()}
dw = {k: v.clone() for k, v in d0.state_dict().items()}
gen = torch.Generator().manual_seed(8)
x = torch.rand(N, 3, 256, 256, generator=gen)
y = (torch.rand(N, 4, 256, 256, generator=gen) > 0.7).float()


def one():
    g = pg.UNet(3, 4, 64, activation='leakyrelu', final_act='softmax', use_dropout=False)
    d = pg.Discriminator(7, 64, n_layers=3)
    g.load_state_dict(gw); d.load_state_dict(dw)
    g.cuda().set_precision(prec); d.cuda().set_precision(prec)
    t = pg.Trainer(g, d, tempfile.mkdtemp())
    t.loss_type = 'weighted_bce'
    t.setup_optimizers(1e-3, 1e-3)
    g.train(); d.train()
    steps = int(os.environ.get('CC_STEPS', '1'))
    extra = {}
    for s_ in range(steps - 1):
        l = t.batch(x, y, train=True)
        if os.environ.get('CC_SYNC') == '1':
            t.flush()
            torch.cuda.synchronize()
            extra[f'step {s_ + 1} G weights'] = g.flat.detach().clone().cpu()
            extra[f'step {s_ + 1} D weights'] = d.flat.detach().clone().cpu()
            extra[f'step {s_ + 1} G grad'] = g.grad_flat.detach().clone().cpu()
            extra[f'step {s_ + 1} D grad'] = d.grad_flat.detach().clone().cpu()
        extra[f'step {s_ + 1} losses'] = torch.tensor([l[k] for k in ('gen', 'gen_loss', 'gdisc', 'discr', 'discf', 'disc')], dtype=torch.float64)
    l = t.batch(x, y, train=True)
    t.flush()
    torch.cuda.synchronize()
    lg = t._last_gen
    return {**extra, 'losses': torch.tensor([l[k] for k in ('gen', 'gen_loss', 'gdisc', 'discr', 'discf', 'disc')], dtype=torch.float64),
            'gen image': lg.t.detach().clone().cpu() if hasattr(lg, 't') else torch.as_tensor(lg).cpu(),
            'G grad': g.grad_flat.detach().clone().cpu(), 'D grad': d.grad_flat.detach().clone().cpu(),
            'G weights': g.flat.detach().clone().cpu(), 'D weights': d.flat.detach().clone().cpu()}


if what == 'save':
    torch.save(one(), path)
    print('saved')
else:
    base = torch.load(path)
    for r in range(reps):
        cur = one()
        msg = []
        for k, a in base.items():
            b = cur[k]
            if not torch.equal(a.contiguous().view(torch.uint8), b.contiguous().view(torch.uint8)):
                af, bf = a.double(), b.double()
                m = ~(torch.isnan(af) | torch.isnan(bf))
                msg.append(f'{k}: rel {float((af[m] - bf[m]).abs().max() / af[m].abs().max()):.1e} ({int(((af != bf) & m).sum())} of {a.numel()})')
        print(f'pid {os.getpid()} rep {r}:', 'all equal' if not msg else ' | '.join(msg), flush=True)
