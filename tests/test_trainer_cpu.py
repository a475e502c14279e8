"""Host logic of the Trainer that needs no GPU: the scalar ReduceLROnPlateau restatement against torch's scheduler
(reference trainer.py:175-178,271-273)."""
import math

import pytest
import torch


def _sequences():
    g = torch.Generator().manual_seed(5)
    noisy = (1.0 + 0.3 * torch.rand(30, generator=g)).tolist()                      # 30 values, never improving on the first few
    falling = [2.0 * 0.97 ** i for i in range(30)]                                  # improves every epoch: no reduction
    stairs = [1.0] * 15 + [0.5] * 15 + [0.49999] * 14 + [0.2] + [0.3] * 40          # plateaus, a sub-threshold "improvement"
    long_flat = [1.0] * 130                                                         # eleven reductions in a row: the eps rule
    edge = [1.0, 1.0 - 1e-4, 1.0 - 1.0001e-4, 1.0 - 2.1e-4] + [1.0] * 26           # around the relative threshold
    return {'noisy': noisy, 'falling': falling, 'stairs': stairs, 'long_flat': long_flat, 'edge': edge}


@pytest.mark.parametrize('name', sorted(_sequences()))
@pytest.mark.parametrize('lr0', [1e-3, 2e-7])
def test_plateau_equals_torch_reduce_on_plateau(name, lr0):
    """Every epoch's learning rate equals that of torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer) with default
    arguments stepped with the same validation-loss sequence (exact float equality: both multiply by 0.1)."""
    from patchgan_amd.trainer import _Plateau
    seq = _sequences()[name]
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=lr0, betas=(0.9, 0.999))
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt)
    mine = _Plateau(lr0)
    for i, v in enumerate(seq):
        sched.step(v)
        got = mine.step(v)
        want = opt.param_groups[0]['lr']
        assert got == want or math.isclose(got, want, rel_tol=1e-15), (name, i, got, want)
    if name == 'falling':
        assert mine.lr == lr0
    if name == 'long_flat' and lr0 == 1e-3:
        assert mine.lr < 1e-8 * 1.2           # reductions stopped once a step would change the rate by less than eps
