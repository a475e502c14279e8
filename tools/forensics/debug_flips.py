"""Debug: LeakyReLU sign flips of d0's output between fp32 kernels and float64 at the cfg2 full-width test's inputs (B = 4)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
import patchgan_amd as pg
from patchgan_amd import engine as E, _lib as L
torch.manual_seed(1234)
g = pg.UNet(3, 1, 64, use_dropout=False, activation='leakyrelu', final_act='sigmoid')
d = pg.Discriminator(4, 64, n_layers=3)
gen = torch.Generator().manual_seed(7)
x = torch.rand(4, 3, 256, 256, generator=gen)
y = (torch.rand(4, 1, 256, 256, generator=gen) > 0.7).float()
W = d.state_dict()['model.0.weight'].cuda()
b = d.state_dict()['model.0.bias'].cuda()
inp = torch.cat((x, y), 1).cuda()
lin64 = F.conv2d(inp.double(), W.double(), b.double(), stride=2, padding=1)
lin32 = F.conv2d(inp, W, b, stride=2, padding=1)
xin = E.View.alloc(4, 256, 256, 4, 'cuda').from_nchw(inp)
out = E.View.alloc(4, 128, 128, 64, 'cuda')
op = E.ConvOp(4, 256, 256, 64, 4, 2, 0)
P = W.permute(2, 3, 0, 1).contiguous().reshape(-1)
op.big2small(xin, P, 0, b, 0, out, L.ACT_NONE)
torch.cuda.synchronize()
hip = out.t.view(4, 128, 128, 64).permute(0, 3, 1, 2)
print(op.describe(0)[0])
print('flips hip vs f64', ((hip > 0) != (lin64 > 0)).sum().item(), ' torch-fp32 vs f64', ((lin32 > 0) != (lin64 > 0)).sum().item(),
      ' elements', lin64.numel(), ' |lin64| < 1e-6:', (lin64.abs() < 1e-6).sum().item())

# the discriminator forward of the training step (2N = 8: real | fake) through the engine
g.cuda(); d.cuda(); g.train(); d.train()
with torch.no_grad():
    fake_y = g(x.cuda())
din_nchw = torch.cat((torch.cat((x.cuda(), y.cuda()), 1), torch.cat((x.cuda(), fake_y), 1)), 0)
din = E.View.alloc(8, 256, 256, 4, 'cuda').from_nchw(din_nchw)
for rep in range(3):
    c = d.engine.forward(d.flat, din)
    torch.cuda.synchronize()
    t0 = c.t[0].t.view(8, 128, 128, 64).permute(0, 3, 1, 2).double()
    want = F.leaky_relu(F.conv2d(din_nchw.double(), W.double(), b.double(), stride=2, padding=1), 0.2)
    err = (t0 - want).abs()
    print('engine d0 output: max abs err', err.max().item(), 'count > 1e-5', (err > 1e-5).sum().item(), 'sign flips', ((t0 > 0) != (want > 0)).sum().item())
