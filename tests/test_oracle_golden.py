"""Pins the CPU oracle (oracle/patchgan_oracle.py) to fixtures generated from the
reference itself (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import patchgan_oracle as O
from tests.golden_util import Golden, CONFIG_NAMES, WIDE_NAMES, LOSS_KEYS, probe


def make_trainer(gold):
    c = gold.cfg
    return O.OracleTrainer(gold.weights('g0'), gold.weights('d0'), activation=c['activation'],
                           final_act=c['final_act'], n_layers=c['n_layers'], norm=c['norm'],
                           loss_type=c['loss_type'], seg_alpha=200, gen_lr=1e-3, dsc_lr=1e-3)


@pytest.mark.parametrize('name', CONFIG_NAMES)
def test_weight_shapes(name):
    gold = Golden(name)
    c = gold.cfg
    gs = O.unet_weight_shapes(c['in_nc'], c['out_nc'], c['nf'])
    ds = O.disc_weight_shapes(c['in_nc'] + c['out_nc'], c['ndf'], c['n_layers'], c['norm'])
    gw, dw = gold.weights('g0'), gold.weights('d0')
    assert list(gs) == list(gw) and all(tuple(gw[k].shape) == gs[k] for k in gs)
    assert list(ds) == list(dw) and all(tuple(dw[k].shape) == ds[k] for k in ds)


@pytest.mark.parametrize('name', CONFIG_NAMES)
def test_forward_probes(name):
    gold = Golden(name)
    c = gold.cfg
    x, y = gold.inputs()
    gw, dw = gold.weights('g0'), gold.weights('d0')
    probes = {}
    with torch.no_grad():
        gen, hid = O.unet_forward(gw, x, c['activation'], c['final_act'], return_hidden=True, probes=probes)
        dfake = O.disc_forward(dw, torch.cat((x, gen), 1), c['n_layers'], c['norm'])
        dreal = O.disc_forward(dw, torch.cat((x, y), 1), c['n_layers'], c['norm'])
    want = gold.probes('fwd')
    assert tuple(want['disc_shape']) == tuple(dfake.shape)
    for k, v in probes.items():
        np.testing.assert_allclose(probe(v), want[k], rtol=1e-6, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(probe(hid), want['hidden'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(probe(dfake), want['disc_fake'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(probe(dreal), want['disc_real'], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize('name', CONFIG_NAMES)
def test_loss_curve_and_grads(name):
    gold = Golden(name)
    x, y = gold.inputs()
    t = make_trainer(gold)
    ev = t.batch(x, y, train=False)
    np.testing.assert_allclose([ev[k] for k in LOSS_KEYS], gold.z['eval_losses'], rtol=1e-6)
    curve = []
    for s in range(gold.nsteps):
        l = t.batch(x, y, train=True)
        curve.append([l[k] for k in LOSS_KEYS])
        if s == 0:
            for k, want in gold.probes('ggrad1').items():
                got = probe(t.last['g_grads'][k])
                np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-6 * max(abs(want[1]), 1e-12), err_msg=k)
            for k, want in gold.probes('dgrad1').items():
                got = probe(t.last['d_grads'][k])
                np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-6 * max(abs(want[1]), 1e-12), err_msg=k)
            for k, want in gold.probes('g1').items():
                np.testing.assert_allclose(probe(t.gw[k]), want, rtol=1e-5, atol=1e-7, err_msg=k)
    curve = np.array(curve)
    # the oracle uses the same torch kernels as the reference: expect (near) bit equality
    np.testing.assert_allclose(curve, gold.z['losses'], rtol=1e-5, atol=1e-5)
    for k, want in gold.probes('g10').items():
        np.testing.assert_allclose(probe(t.gw[k]), want, rtol=1e-4, atol=1e-6, err_msg=k)
    for k, want in gold.probes('d10').items():
        np.testing.assert_allclose(probe(t.dw[k]), want, rtol=1e-4, atol=1e-6, err_msg=k)


@pytest.mark.parametrize('name', WIDE_NAMES)
def test_wide_benchmark_configs_vs_reference(name):
    """The oracle at the BENCHMARK configurations against fixtures the reference itself produced at full width (cfg2: nf = ndf = 64,
    bs 16, 10 steps; cfg1: the COCO yaml's hyper-parameters, 10 steps; cfg4's shape at B = 2, 4 steps): the seeded default init is
    the reference's bit for bit (weight probes), and the evaluation losses, every loss of every step, the step-1 gradient probes and
    the final weight probes agree.  The oracle runs the same torch-CPU kernels as the reference: with the thread count the fixtures
    were made with (8) the curves are BIT-EQUAL (asserted); with another count oneDNN may sum in another order (1e-5 then)."""
    gold = Golden(name)
    c = gold.cfg
    g, d = gold.seeded_modules()                     # asserts the weight probes
    gw = {k: v.clone() for k, v in g.state_dict().items()}
    dw = {k: v.clone() for k, v in d.state_dict().items()}
    x, y = gold.inputs()
    t = O.OracleTrainer(gw, dw, activation=c['activation'], final_act=c['final_act'], n_layers=c['n_layers'], norm=c['norm'],
                        loss_type=c['loss_type'], seg_alpha=200, gen_lr=1e-3, dsc_lr=1e-3)
    same_threads = torch.get_num_threads() == int(gold.z['threads'][0])
    ev = t.batch(x, y, train=False)
    np.testing.assert_allclose([ev[k] for k in LOSS_KEYS], gold.z['eval_losses'], rtol=1e-6)
    curve = []
    for s in range(gold.nsteps):
        l = t.batch(x, y, train=True)
        curve.append([l[k] for k in LOSS_KEYS])
        if s == 0:
            for kind, grads in (('ggrad1', t.last['g_grads']), ('dgrad1', t.last['d_grads'])):
                for k, want in gold.probes(kind).items():
                    np.testing.assert_allclose(probe(grads[k]), want, rtol=2e-5, atol=1e-6 * max(abs(want[1]), 1e-12), err_msg=k)
    curve = np.array(curve)
    if same_threads:
        assert np.array_equal(curve, gold.z['losses']), np.abs(curve - gold.z['losses']).max()
    np.testing.assert_allclose(curve, gold.z['losses'], rtol=1e-5, atol=1e-5)
    for k, want in gold.probes('g10').items():
        np.testing.assert_allclose(probe(t.gw[k]), want, rtol=1e-4, atol=1e-6, err_msg=k)
    for k, want in gold.probes('d10').items():
        np.testing.assert_allclose(probe(t.dw[k]), want, rtol=1e-4, atol=1e-6, err_msg=k)


def test_lr_schedule():
    import os
    from tests.golden_util import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, 'train_driver.npz'))
    # LR in the optimizer after `e` epochs == LR that epoch e+1 would print
    for e in range(1, 7):
        seq = O.exponential_lr_sequence(1e-3, 0.9, e + 1, decay_freq=2)
        assert abs(seq[e] - z['lr_after_epochs'][e - 1, 0]) < 1e-12
        seq = O.exponential_lr_sequence(2e-3, 0.9, e + 1, decay_freq=2)
        assert abs(seq[e] - z['lr_after_epochs'][e - 1, 1]) < 1e-12
    assert abs(O.resume_lr(1e-3, 0.9, 7, 2) - 1e-3 * 0.9 ** 3) < 1e-15
