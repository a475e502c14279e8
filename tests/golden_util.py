"""Helpers shared by the parity tests: fixture loading, probes, input recipe."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
LOSS_KEYS = ['gen', 'gen_loss', 'gdisc', 'discr', 'discf', 'disc']
CONFIG_NAMES = ['a_lrelu_tversky', 'b_tanh_wbce_norm', 'c_relu_mae_l5', 'd_softmax_tversky', 'e_wbce_c1']
# the BENCHMARK configurations at full width, generated from the reference itself (tests/golden/make_golden.py WIDE_CONFIGS): cfg2 (nf =
# ndf = 64, bs 16, 10 steps), cfg1 (COCO yaml hyper-parameters, bs 4, 10 steps), cfg4's shape (512 x 512, 4 classes, B = 2, 4 steps).
# They hold probes of the initial weights instead of the weights: those are torch's default init under the recorded seed.
WIDE_NAMES = ['w_cfg1', 'w_cfg2', 'w_cfg4']


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

    def seeded_modules(self):
        """(generator, discriminator) of this configuration as patchgan_amd modules initialised like the reference's were (torch's
        default init under the recorded seed, G constructed before D), with the golden's weight probes checked tensor by tensor --
        for the wide fixtures, which do not carry the 41.8 M initial weights."""
        import patchgan_amd as pg
        c = self.cfg
        torch.manual_seed(self.model_seed)
        g = pg.UNet(c['in_nc'], c['out_nc'], c['nf'], use_dropout=False, activation=c['activation'], final_act=c['final_act'])
        d = pg.Discriminator(c['in_nc'] + c['out_nc'], c['ndf'], n_layers=c['n_layers'], norm=c['norm'])
        for prefix, net in (('g0', g), ('d0', d)):
            for k, v in net.state_dict().items():
                got, want = probe(v), self.z[f'{prefix}/{k}']
                # the 64 samples bit for bit; the two float64 sums over up to 8.4 M elements to 1e-12 of the abs-sum (their last bits
                # depend on the host's reduction order: thread count, vector width)
                assert np.array_equal(got[2:], want[2:]) and np.abs(got[:2] - want[:2]).max() <= 1e-12 * want[1], \
                    f'{self.name}: initial {prefix}/{k} differs from the reference\'s'
        return g, d

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
