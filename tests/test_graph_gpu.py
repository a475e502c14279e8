"""Trainer.graph: the steady-state training step replayed from a captured hipGraph must be the SAME computation as the launch-by-launch
step -- bit-identical losses and weights, with Adam's step count and learning rate moving on inside the replays (pg_adam_step_dev) --
and must fall back to launch by launch where it cannot apply (dropout, evaluation passes, a changed batch shape).  Needs an MI355X."""
import numpy as np
import pytest
import torch

from tests.golden_util import LOSS_KEYS

pytestmark = pytest.mark.gpu


def _run(tmp_path, graph, precision, steps, nf=16, lr_change_at=None, use_dropout=False, tag=''):
    import patchgan_amd as pg
    torch.manual_seed(99)
    g = pg.UNet(3, 1, nf, use_dropout=use_dropout, activation='leakyrelu', final_act='sigmoid')
    d = pg.Discriminator(4, nf, n_layers=3)
    g.cuda()
    d.cuda()
    if precision == 'bf16':
        g.set_precision('bf16')
        d.set_precision('bf16')
    t = pg.Trainer(g, d, str(tmp_path / f'ck_{graph}_{precision}{tag}'))
    t.graph = graph
    t.setup_optimizers(1e-3, 2e-3)
    g.train()
    d.train()
    gen = torch.Generator().manual_seed(5)
    rows, used = [], []
    for s in range(steps):
        x = torch.rand(2, 3, 256, 256, generator=gen)          # new inputs every step: the replay must read the CURRENT batch
        y = (torch.rand(2, 1, 256, 256, generator=gen) > 0.7).float()
        if lr_change_at is not None and s == lr_change_at:
            t.gen_lr, t.dsc_lr = 5e-4, 1e-4                    # ExponentialLR / plateau steps change these between epochs
        l = t.batch(x, y, train=True)
        rows.append([float(l[k]) for k in LOSS_KEYS])
        used.append(t.graph_captured())
    torch.cuda.synchronize()
    return np.array(rows), g.flat.cpu().numpy().copy(), d.flat.cpu().numpy().copy(), used, t


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_graph_replay_is_bit_identical_to_the_eager_step(tmp_path, precision):
    nf = 32 if precision == 'bf16' else 16
    a = _run(tmp_path, False, precision, 9, nf=nf, lr_change_at=6)
    b = _run(tmp_path, True, precision, 9, nf=nf, lr_change_at=6)
    assert not any(a[3]) and b[3][:3] == [False] * 3 and all(b[3][3:]), (a[3], b[3])     # captured at the 4th step of its kind
    assert np.array_equal(a[0], b[0]), (a[0] - b[0])
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert b[4]._t_g == 9 and b[4]._t_d == 9


def test_graph_falls_back_where_it_cannot_apply(tmp_path):
    import patchgan_amd as pg
    # dropout on: per-layer seeds are launch arguments that change every step -> launch by launch, masks differ from step to step
    rows, _, _, used, t = _run(tmp_path, True, 'fp32', 6, use_dropout=True, tag='drop')
    assert not any(used)
    # a captured trainer: an evaluation pass and a different batch size run eagerly and leave the captured step usable
    rows, gw, dw, used, t = _run(tmp_path, True, 'fp32', 5, tag='mix')
    assert used[-1]
    gen = torch.Generator().manual_seed(6)
    x = torch.rand(2, 3, 256, 256, generator=gen)
    y = (torch.rand(2, 1, 256, 256, generator=gen) > 0.7).float()
    ev = t.batch(x, y, train=False)
    assert np.isfinite(ev['gen'])
    assert np.array_equal(t.generator.flat.cpu().numpy(), gw)            # no update in an evaluation pass
    l3 = t.batch(torch.cat([x, x[:1]]), torch.cat([y, y[:1]]), train=True)        # N = 3: eager (its kind has not been seen)
    assert np.isfinite(l3['disc']) and t._t_g == 6
    l2 = t.batch(x, y, train=True)                                              # back on the captured graph
    assert np.isfinite(l2['disc']) and t._t_g == 7
    # against an eager trainer that saw the same sequence
    rows_e, _, _, _, te = _run(tmp_path, False, 'fp32', 5, tag='mix_e')
    te.batch(x, y, train=False)
    te.batch(torch.cat([x, x[:1]]), torch.cat([y, y[:1]]), train=True)
    l2e = te.batch(x, y, train=True)
    assert [float(l2[k]) for k in LOSS_KEYS] == [float(l2e[k]) for k in LOSS_KEYS]
    torch.cuda.synchronize()
    assert np.array_equal(t.generator.flat.cpu().numpy(), te.generator.flat.cpu().numpy())


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_auto_policy_both_outcomes_are_the_same_computation(tmp_path, precision, monkeypatch):
    """Trainer.graph = 'auto': the last launch-by-launch warm step is timed on device and host; a device-bound step stays launch by
    launch with the weight-gradient chain of each backward pass on a second stream (fp32 networks), a launch-bound one is captured.
    Both outcomes, forced through AUTO_RATIO, against the plain one-stream step: bit-identical losses and weights."""
    import patchgan_amd as pg
    from patchgan_amd import engine as E
    nf = 32 if precision == 'bf16' else 16
    ref = _run(tmp_path, False, precision, 9, nf=nf, lr_change_at=6, tag='ref')
    # device-bound by decree: never captured; from the 5th step of its kind on, two streams (fp32)
    monkeypatch.setattr(pg.Trainer, 'AUTO_RATIO', 0.0)
    sides = []
    orig = E._side_begin

    def spy(allow):
        orig(allow)
        sides.append(bool(E._SIDE['allow']))
    monkeypatch.setattr(E, '_side_begin', spy)
    forks = []
    enter = E.on_side.__enter__

    def spy_enter(self):
        forks.append(1)
        return enter(self)
    monkeypatch.setattr(E.on_side, '__enter__', spy_enter)
    a = _run(tmp_path, 'auto', precision, 9, nf=nf, lr_change_at=6, tag='eager')
    monkeypatch.setattr(E, '_side_begin', orig)
    monkeypatch.setattr(E.on_side, '__enter__', enter)
    assert len(forks) >= 4, forks          # the discriminator step's forward went to the second stream (both precisions), steps 5..9
    assert not any(a[3]) and a[4].graph_decided() and a[4].step_times is not None and a[4].step_times[0] > 0
    assert any(sides) == (precision == 'fp32'), sides          # weight gradients on the second stream: fp32 networks only
    assert np.array_equal(ref[0], a[0]) and np.array_equal(ref[1], a[1]) and np.array_equal(ref[2], a[2])
    assert not E._SIDE['keep'] and not E._SIDE['enabled']
    # launch-bound by decree: captured at the 4th step of its kind, as with graph = True
    monkeypatch.setattr(pg.Trainer, 'AUTO_RATIO', 1e9)
    b = _run(tmp_path, 'auto', precision, 9, nf=nf, lr_change_at=6, tag='graph')
    assert b[3][:3] == [False] * 3 and all(b[3][3:]), b[3]
    assert np.array_equal(ref[0], b[0]) and np.array_equal(ref[1], b[1]) and np.array_equal(ref[2], b[2])


def test_two_stream_step_at_full_width_is_bit_identical(tmp_path, monkeypatch):
    """nf = ndf = 64 (the benchmark's widths: every Winograd path, the decoder's backward call as its two halves): the two-stream
    launch-by-launch step against the one-stream one, 6 steps from the same start -- identical losses and weights, bit for bit."""
    import patchgan_amd as pg
    monkeypatch.setattr(pg.Trainer, 'AUTO_RATIO', 0.0)
    ref = _run(tmp_path, False, 'fp32', 6, nf=64, tag='w_ref')
    two = _run(tmp_path, 'auto', 'fp32', 6, nf=64, tag='w_two')
    assert two[4].graph_decided() and not any(two[3])
    assert np.array_equal(ref[0], two[0]) and np.array_equal(ref[1], two[1]) and np.array_equal(ref[2], two[2])
