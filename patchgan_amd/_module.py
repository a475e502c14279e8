"""Shared nn.Module plumbing: parameters that are strided views of one flat packed buffer."""
import torch
from torch import nn

from . import engine as E


class FlatParamModule(nn.Module):
    """Holds `self.flat` (packed weights, see engine.py) and registers one nn.Parameter per state_dict key of
    the reference, each a strided view of the flat buffer -- so state_dict()/load_state_dict()/parameters()/
    to()/apply() behave like the reference's modules while kernels, Adam and all-reduce see one buffer."""

    # set by a Trainer: completes an update of this module that is still in flight (Trainer.flush) before its weights or gradients are
    # read through the module's own surface (state_dict, parameters, forward, .to()).  `flat` / `grad_flat` are the raw buffers the
    # engines work on: whoever reads those directly calls Trainer.flush() first.  So does whoever keeps a reference to a CHILD's parameter
    # (module.model[i].weight, an optimizer built earlier) across batch() calls: attribute access on a child is not intercepted.
    _access_hook = None

    def _pre_access(self):
        hook = self._access_hook
        if hook is not None:
            hook()

    def state_dict(self, *args, **kwargs):
        self._pre_access()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._pre_access()
        return super().load_state_dict(*args, **kwargs)

    def parameters(self, *args, **kwargs):
        self._pre_access()
        return super().parameters(*args, **kwargs)

    def named_parameters(self, *args, **kwargs):
        self._pre_access()
        return super().named_parameters(*args, **kwargs)

    def get_parameter(self, target):
        self._pre_access()
        return super().get_parameter(target)

    def apply(self, fn):
        self._pre_access()
        return super().apply(fn)

    def _init_flat(self, layers, nparams):
        self._layers = layers
        flat = torch.zeros(nparams, dtype=torch.float32)
        E.default_init_(flat, layers)
        self._bind(flat)

    def _bind(self, flat):
        object.__setattr__(self, 'flat', flat)
        object.__setattr__(self, 'grad_flat', None)
        views = E.torch_views(flat, self._layers)
        for key, v in views.items():
            mod, leaf = self._owner(key)
            if leaf in mod._parameters and mod._parameters[leaf] is not None:
                mod._parameters[leaf].data = v
            else:
                mod.register_parameter(leaf, nn.Parameter(v))

    def _owner(self, key):
        parts = key.split('.')
        mod = self
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        return mod, parts[-1]

    def _apply(self, fn, recurse=True):
        # .to()/.cuda()/.cpu(): move the flat buffer, then re-point every parameter at its view of it
        self._pre_access()
        new_flat = fn(self.flat.detach())
        if new_flat.dtype != torch.float32:
            raise TypeError("patchgan_amd networks are fp32 only")
        self._bind(new_flat.contiguous())
        for p in self.parameters():
            p.grad = None
        return self

    def set_precision(self, precision, bf16_storage=True):
        """'fp32' (default: exact fp32 MFMA, the parity path) or 'bf16': bf16 MFMA with fp32 accumulation, and -- unless
        bf16_storage=False or a channel count rules it out -- the interior activations and their gradients stored as bf16 in
        HBM.  Master weights, weight gradients, InstanceNorm statistics, losses and Adam stay fp32 either way.  Returns self."""
        from . import _lib as L
        algo = {'fp32': L.ALGO_AUTO, 'bf16': L.ALGO_BF16}[precision]
        self.engine.algo = algo | (self.engine.algo & ~L.ALGO_MASK)
        self.engine._ops = {}
        self.engine._sok = {}
        self.engine.clear_weight_caches()
        # bf16 activation storage wherever every interior tensor can take it (channel counts % 4 == 0 and >= 32); otherwise the
        # bf16 kernels keep reading fp32 tensors and rounding them in flight
        self.engine.act_bf = bool(precision == 'bf16' and bf16_storage and self.engine.bf16_storage_ok())
        self.precision = precision
        return self

    def set_tuning(self, bits):
        """Per-call PG_TUNE_* bits (patchgan_amd._lib.TUNE_*) for every convolution of this network: overrides the kernel
        selection heuristics (e.g. TUNE_WINO2_ALL forces the polyphase Winograd path wherever the geometry allows,
        TUNE_WINO_OFF gives the exact implicit GEMM everywhere).  Returns self."""
        from . import _lib as L
        self.engine.algo = (self.engine.algo & L.ALGO_MASK) | int(bits)
        self.engine._ops = {}
        self.engine._sok = {}
        self.engine.clear_weight_caches()
        return self

    def ensure_grad_flat(self):
        """Flat gradient buffer in the packed layout; every parameter's .grad is a view of it."""
        if self.grad_flat is None or self.grad_flat.device != self.flat.device:
            g = torch.zeros_like(self.flat)
            object.__setattr__(self, 'grad_flat', g)
            for key, v in E.torch_views(g, self._layers).items():
                mod, leaf = self._owner(key)
                mod._parameters[leaf].grad = v
        return self.grad_flat

    def state_dict_contiguous(self):
        """state_dict() with every tensor in the reference's contiguous NCHW layout (what save() writes)."""
        return {k: v.detach().contiguous().clone() for k, v in self.state_dict().items()}
