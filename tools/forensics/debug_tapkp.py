"""Debug: k_b2s_tapkp vs float64 on the d0 geometry at several batch sizes; prints max abs error and its location."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from patchgan_amd import engine as E, _lib as L
dev = 'cuda'
for N in (4, 8, 16):
    for Cb, ld in ((4, 4), (3, 4)):
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn(N, 256, 256, ld, device=dev, generator=g)
        W = torch.randn(64, Cb, 4, 4, device=dev, generator=g) * 0.1
        b = torch.randn(64, device=dev, generator=g)
        big = E.View(x.reshape(-1), 0, ld, N, 256, 256, Cb)
        out = E.View.alloc(N, 128, 128, 64, dev)
        op = E.ConvOp(N, 256, 256, 64, Cb, 2, 0)
        P = W.permute(2, 3, 0, 1).contiguous().reshape(-1)
        op.big2small(big, P, 0, b, 0, out, L.ACT_LEAKY)
        torch.cuda.synchronize()
        want = F.leaky_relu(F.conv2d(x[..., :Cb].permute(0, 3, 1, 2).double(), W.double(), b.double(), stride=2, padding=1), 0.2)
        got = out.t.view(N, 128, 128, 64).permute(0, 3, 1, 2).double()
        err = (got - want).abs()
        i = err.argmax().item()
        idx = []
        for d in reversed(want.shape):
            idx.append(i % d); i //= d
        print(op.describe(0)[0], 'N', N, 'Cb', Cb, 'max abs err', err.max().item(), 'at (n,c,h,w)', idx[::-1], 'nan', torch.isnan(got).sum().item(),
              'count>1e-4', (err > 1e-4).sum().item())
