"""UNet generator with the reference's constructor / forward / state_dict surface (patchgan/unet.py:75-134),
executed by hand-written gfx950 kernels (patchgan_amd.engine.GeneratorEngine)."""
import torch
from torch import nn

from . import engine as E
from ._module import FlatParamModule
from .transfer import Transferable


class _UNetFn(torch.autograd.Function):
    """Whole-network autograd node: forward = engine forward, backward = engine backward."""

    @staticmethod
    def forward(ctx, module, x, return_hidden, *params):
        eng = module.engine
        N, C, H, W = x.shape
        dev = module.flat.device
        xin = E.View.alloc(N, H, W, C, dev).from_nchw(x.to(device=dev, dtype=torch.float32))
        gen = E.View.alloc(N, H, W, eng.output_nc, dev)
        seed = module._next_seed() if module.training else 0
        c = eng.forward(module.flat, xin, gen, module.training, seed)
        ctx.module, ctx.c = module, c
        ctx.need_dx = x.requires_grad
        out = gen.to_nchw()
        if return_hidden:
            hidden = c.hidden.to_nchw()
            ctx.mark_non_differentiable(hidden)   # side output: gradients flow through `out` only
            return out, hidden
        return out

    @staticmethod
    def backward(ctx, gout, *unused):
        module, c = ctx.module, ctx.c
        eng = module.engine
        dev = module.flat.device
        g = E.View.alloc(c.N, c.H, c.W, eng.output_nc, dev).from_nchw(gout.to(torch.float32))
        gflat = torch.zeros_like(module.flat)
        dx = eng.backward(module.flat, gflat, c, g, None, need_dx=ctx.need_dx)
        views = E.torch_views(gflat, eng.layers)
        grads = tuple(views[k] for k in module._param_keys)
        return (None, dx.to_nchw() if dx is not None else None, None) + grads


class UNet(FlatParamModule, Transferable):
    """UNet(input_nc, output_nc, nf=64, norm_layer=InstanceNorm2d, use_dropout=False, activation='tanh',
    final_act='softmax') -- reference unet.py:76-78.  Only nn.InstanceNorm2d is supported as norm_layer."""

    def __init__(self, input_nc, output_nc, nf=64, norm_layer=nn.InstanceNorm2d, use_dropout=False,
                 activation='tanh', final_act='softmax'):
        super().__init__()
        if norm_layer is not nn.InstanceNorm2d:
            raise NotImplementedError("patchgan_amd.UNet implements nn.InstanceNorm2d blocks only")
        self.engine = E.GeneratorEngine(input_nc, output_nc, nf, activation, final_act, use_dropout)
        self._param_keys = [l.key for l in self.engine.layers]
        self._seed_base = int(torch.initial_seed()) & 0xFFFFFFFF
        self._calls = 0
        self._init_flat(self.engine.layers, self.engine.nparams)

    def _next_seed(self):
        self._calls += 1
        return E._mix_seed(self._seed_base, self._calls)

    def forward(self, x, return_hidden=False):
        params = [self.get_parameter(k) for k in self._param_keys]
        return _UNetFn.apply(self, x, return_hidden, *params)
