"""Debug: step-1 discriminator gradients vs the float64 oracle at cfg2 width (B = 4), per parameter."""
import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import patchgan_amd as pg
from oracle import patchgan_oracle as O
torch.manual_seed(1234)
g = pg.UNet(3, 1, 64, use_dropout=False, activation='leakyrelu', final_act='sigmoid')
d = pg.Discriminator(4, 64, n_layers=3)
gw = {k: v.clone() for k, v in g.state_dict().items()}
dw = {k: v.clone() for k, v in d.state_dict().items()}
gen = torch.Generator().manual_seed(7)
x = torch.rand(4, 3, 256, 256, generator=gen)
y = (torch.rand(4, 1, 256, 256, generator=gen) > 0.7).float()
kw = dict(activation='leakyrelu', final_act='sigmoid', n_layers=3, norm=False, loss_type='tversky')
o64 = O.OracleTrainer({k: v.cuda() for k, v in gw.items()}, {k: v.cuda() for k, v in dw.items()}, dtype=torch.float64, **kw)
o64.batch(x.cuda(), y.cuda(), train=True)
t = pg.Trainer(g.cuda(), d.cuda(), tempfile.mkdtemp())
t.setup_optimizers(1e-3, 1e-3)
g.train(); d.train()
t.batch(x, y, train=True)
for k, p in d.named_parameters():
    a, b = p.grad.double().cpu(), o64.last['d_grads'][k].double().cpu()
    print(f'{k:16s} rel max {((a-b).abs().max()/b.abs().max()).item():.2e}  |want|max {b.abs().max().item():.3e}')
a, b = d.get_parameter('model.0.bias').grad.double().cpu(), o64.last['d_grads']['model.0.bias'].double().cpu()
print('bias grad got ', a[:8].tolist())
print('bias grad want', b[:8].tolist())
