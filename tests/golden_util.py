"""Helpers shared by the parity tests: fixture loading, probes, input recipe."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
LOSS_KEYS = ['gen', 'gen_loss', 'gdisc', 'discr', 'discf', 'disc']
CONFIG_NAMES = ['a_lrelu_tversky', 'b_tanh_wbce_norm', 'c_relu_mae_l5', 'd_softmax_tversky', 'e_wbce_c1']


def _conv(v):
    if v in ('True', 'False'):
        return v == 'True'
    try:
        return int(v)
    except ValueError:
        return v


class Golden:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
        self.cfg = {k: _conv(v) for k, v in zip(self.z['cfg_keys'], self.z['cfg_vals'])}
        self.model_seed, self.data_seed, self.nsteps, self.nsamp = [int(v) for v in self.z['meta']]

    def weights(self, prefix):
        """prefix 'g0' / 'd0' -> {state_dict key: float32 tensor}"""
        p = prefix + '/'
        return {k[len(p):]: torch.from_numpy(self.z[k].copy()) for k in self.z.files if k.startswith(p)}

    def probes(self, prefix):
        p = prefix + '/'
        return {k[len(p):]: self.z[k] for k in self.z.files if k.startswith(p)}

    def inputs(self):
        c = self.cfg
        g = torch.Generator().manual_seed(self.data_seed)
        x = torch.rand(c['B'], c['in_nc'], c['size'], c['size'], generator=g)
        y = (torch.rand(c['B'], c['out_nc'], c['size'], c['size'], generator=g) > 0.7).float()
        return x, y


def probe(t, nsamp=64):
    """Same recipe as tests/golden/make_golden.py:probe."""
    f = t.detach().double().flatten().cpu()
    n = f.numel()
    idx = (torch.arange(nsamp, dtype=torch.int64) * 2654435761 % n)
    return np.concatenate([[f.sum().item(), f.abs().sum().item()], f[idx].numpy()])


def probe_close(got, want, rtol, atol_scale=1.0):
    """Compare a probe: sums relative to abs-sum, samples relative to the mean magnitude."""
    got = np.asarray(got)
    want = np.asarray(want)
    scale = max(abs(want[1]), 1e-30)
    err_sum = abs(got[0] - want[0]) / scale
    err_abs = abs(got[1] - want[1]) / scale
    mag = max(np.abs(want[2:]).max(), 1e-30)
    err_s = np.abs(got[2:] - want[2:]).max() / mag
    return max(err_sum, err_abs, err_s) <= rtol * atol_scale, (err_sum, err_abs, err_s)
