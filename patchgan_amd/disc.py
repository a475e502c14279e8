"""N-layer PatchGAN discriminator with the reference's surface (patchgan/disc.py:5-51) on gfx950 kernels."""
import torch
from torch import nn

from . import engine as E
from ._module import FlatParamModule
from .transfer import Transferable


class _DiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, *params):
        eng = module.engine
        N, C, H, W = x.shape
        dev = module.flat.device
        din = E.View.alloc(N, H, W, C, dev).from_nchw(x.to(device=dev, dtype=torch.float32))
        c = eng.forward(module.flat, din)
        ctx.module, ctx.c = module, c
        ctx.need_dx = x.requires_grad
        return c.out.to_nchw()

    @staticmethod
    def backward(ctx, gout):
        module, c = ctx.module, ctx.c
        eng = module.engine
        o = c.out
        g = E.View.alloc(o.N, o.H, o.W, 1, module.flat.device).from_nchw(gout.to(torch.float32))
        gflat = torch.zeros_like(module.flat)
        dx = eng.backward(module.flat, gflat, c, g, need_wgrad=True, need_dx=ctx.need_dx)
        views = E.torch_views(gflat, eng.layers)
        grads = tuple(views[k] for k in module._param_keys)
        return (None, dx.to_nchw() if dx is not None else None) + grads


class Discriminator(FlatParamModule, Transferable):
    """Discriminator(input_nc, ndf=64, n_layers=3, norm=False, norm_layer=InstanceNorm2d) -- reference disc.py:8."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm=False, norm_layer=nn.InstanceNorm2d):
        super().__init__()
        if norm and norm_layer is not nn.InstanceNorm2d:
            raise NotImplementedError("patchgan_amd.Discriminator implements nn.InstanceNorm2d only")
        self.engine = E.DiscriminatorEngine(input_nc, ndf, n_layers, norm)
        self._param_keys = []
        for l in self.engine.layers:
            self._param_keys.append(l.key)
            if l.bias_key is not None:
                self._param_keys.append(l.bias_key)
        self._init_flat(self.engine.layers, self.engine.nparams)

    def forward(self, input):
        self._pre_access()
        params = [self.get_parameter(k) for k in self._param_keys]
        return _DiscFn.apply(self, input, *params)
