"""Trainer.graph: the steady-state training step replayed from a captured hipGraph must be the SAME computation as the launch-by-launch
step -- bit-identical losses and weights, with Adam's step count and learning rate moving on inside the replays (pg_adam_step_dev) --
and must fall back to launch by launch where it cannot apply (dropout, evaluation passes, a changed batch shape).  Needs an MI355X."""
import numpy as np
import pytest
import torch

from tests.golden_util import LOSS_KEYS

pytestmark = pytest.mark.gpu


def _run(tmp_path, graph, precision, steps, nf=16, lr_change_at=None, use_dropout=False, tag='', two_streams=None, eval_at=()):
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
    t.two_streams = two_streams
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
        if s in eval_at:                                       # an evaluation pass in between (its own kind of step)
            g.eval(), d.eval()
            ev = t.batch(x, y, train=False)
            rows.append([float(ev[k]) for k in LOSS_KEYS])
            g.train(), d.train()
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
    launch with the weight-gradient chain of each backward pass on a second stream, a launch-bound one is captured.
    Both outcomes, decreed through AUTO_FORCE, against the plain one-stream step: bit-identical losses and weights."""
    import patchgan_amd as pg
    from patchgan_amd import engine as E
    nf = 32 if precision == 'bf16' else 16
    ref = _run(tmp_path, False, precision, 9, nf=nf, lr_change_at=6, tag='ref')
    # device-bound by decree: never captured; from the 5th step of its kind on, two streams
    monkeypatch.setattr(pg.Trainer, 'AUTO_FORCE', 'eager2')
    sides = []
    orig = E._side_begin

    def spy(allow):
        orig(allow)
        sides.append(bool(E.cur_exec().allow))
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
    assert not any(a[3]) and a[4].graph_decided() and a[4].decided_modes() == ['eager2']
    assert any(sides), sides          # weight gradients on the second stream (fp32 networks since round 4, bf16 ones since round 6)
    assert np.array_equal(ref[0], a[0]) and np.array_equal(ref[1], a[1]) and np.array_equal(ref[2], a[2])
    ex = a[4]._exec
    assert ex.pending and a[4]._deferred is not None          # the last step's discriminator backward pass is still on the second stream ...
    a[4].flush()                                              # ... until something needs the discriminator's weights
    assert a[4]._deferred is None
    assert not ex.keep and not ex.enabled and not ex.pending and ex.device == torch.device('cuda', torch.cuda.current_device())
    assert getattr(E._TLS, 'cur', None) is None          # the trainer's execution state is current only inside batch()
    # launch-bound by decree: captured at the 4th step of its kind, as with graph = True
    monkeypatch.setattr(pg.Trainer, 'AUTO_FORCE', 'graph')
    b = _run(tmp_path, 'auto', precision, 9, nf=nf, lr_change_at=6, tag='graph')
    assert b[3][:3] == [False] * 3 and all(b[3][3:]), b[3]
    assert np.array_equal(ref[0], b[0]) and np.array_equal(ref[1], b[1]) and np.array_equal(ref[2], b[2])


def test_two_stream_step_at_full_width_is_bit_identical(tmp_path, monkeypatch):
    """nf = ndf = 64 (the benchmark's widths: every Winograd path, the decoder's backward call as its two halves): the two-stream
    launch-by-launch step against the one-stream one, 6 steps from the same start -- identical losses and weights, bit for bit."""
    import patchgan_amd as pg
    monkeypatch.setattr(pg.Trainer, 'AUTO_FORCE', 'eager2')
    ref = _run(tmp_path, False, 'fp32', 6, nf=64, tag='w_ref')
    two = _run(tmp_path, 'auto', 'fp32', 6, nf=64, tag='w_two')
    assert two[4].graph_decided() and not any(two[3])
    assert np.array_equal(ref[0], two[0]) and np.array_equal(ref[1], two[1]) and np.array_equal(ref[2], two[2])


def test_dropout_and_eval_steps_take_the_second_stream_and_stay_bit_identical(tmp_path, monkeypatch):
    """The reference CLI's default generator has dropout on (train.py:92, unet.py:26-28,63-65): such a step cannot be captured
    (per-layer seeds are launch arguments) but the two-stream schedule applies to it like to any other -- seeds do not care which
    stream a launch is on.  Same for evaluation passes (the discriminator step's forward on the second stream).  nf = ndf = 64, six
    training steps with an evaluation pass before steps 3 and 5: forced two-stream and 'auto' (device-bound by decree) against the
    one-stream trainer -- identical losses and weights, bit for bit; and the second stream really was used."""
    import patchgan_amd as pg
    from patchgan_amd import engine as E
    ref = _run(tmp_path, False, 'fp32', 6, nf=64, use_dropout=True, tag='d_ref', eval_at=(2, 4))
    assert len(ref[0]) == 8
    forks = []
    enter = E.on_side.__enter__

    def spy_enter(self):
        forks.append(E.cur_exec().enabled)
        return enter(self)
    monkeypatch.setattr(E.on_side, '__enter__', spy_enter)
    two = _run(tmp_path, False, 'fp32', 6, nf=64, use_dropout=True, tag='d_two', two_streams=True, eval_at=(2, 4))
    # every step, the evaluation passes included, forks the discriminator step's forward; every training step also its backward pass
    assert len(forks) == 8 + 6 and all(forks), forks
    assert two[4].launch_mode == 'eager2' and not any(two[3])
    assert np.array_equal(ref[0], two[0]) and np.array_equal(ref[1], two[1]) and np.array_equal(ref[2], two[2])
    monkeypatch.setattr(pg.Trainer, 'AUTO_FORCE', 'eager2')
    del forks[:]
    auto = _run(tmp_path, 'auto', 'fp32', 6, nf=64, use_dropout=True, tag='d_auto', eval_at=(2, 4))
    assert not any(auto[3]) and auto[4].launch_mode == 'eager2' and len(forks) >= 3, forks      # training steps 4..6 (probes on steps 2, 3)
    assert np.array_equal(ref[0], auto[0]) and np.array_equal(ref[1], auto[1]) and np.array_equal(ref[2], auto[2])
    # launch-bound by decree: dropout rules capture out, so the step stays on one stream
    monkeypatch.setattr(pg.Trainer, 'AUTO_FORCE', 'graph')
    one = _run(tmp_path, 'auto', 'fp32', 5, nf=16, use_dropout=True, tag='d_one')
    assert not any(one[3]) and one[4].launch_mode == 'eager1' and one[4].graph_decided()


def test_auto_tournament_measures_every_candidate_and_is_the_same_computation(tmp_path):
    """Trainer.graph = 'auto' without a decree: after one warm step each candidate (one stream, two streams, and -- where the one-stream
    step is close to host-bound -- the captured graph) runs TRIAL_STEPS timed steps, the fastest is kept.  Every candidate is the same
    computation: 36 steps that pass through all of them equal 36 one-stream steps bit for bit.  A step that raises inside the
    tournament does not score (the tournament starts over), and redecide() forgets the outcome."""
    import patchgan_amd as pg
    ref = _run(tmp_path, False, 'fp32', 36, nf=16, tag='t_ref')
    auto = _run(tmp_path, 'auto', 'fp32', 36, nf=16, tag='t_auto')
    t = auto[4]
    assert t.graph_decided() and t.decided_modes()[0] in ('eager1', 'eager2', 'graph')
    ms = {k: v for k, v in t.step_times.items() if k != 'host_enqueue'}
    assert set(ms) == {'eager1', 'eager2', 'graph'} and ms['eager1'] > 0 and ms['eager2'] > 0, t.step_times
    tried = {k: v for k, v in ms.items() if v is not None}
    assert t.decided_modes()[0] == min(tried, key=tried.get), (t.decided_modes(), ms)
    if t.decided_modes()[0] != 'graph':
        assert not t.graph_captured()                  # a losing capture is released
    assert np.array_equal(ref[0], auto[0]) and np.array_equal(ref[1], auto[1]) and np.array_equal(ref[2], auto[2])
    # a raising step inside a tournament
    t.redecide()
    assert not t.graph_decided()
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 256, 256, generator=gen)
    y = (torch.rand(2, 1, 256, 256, generator=gen) > 0.7).float()
    for _ in range(3):
        t.batch(x, y, train=True)
    kind = next(iter(t._kinds.values()))
    assert kind['trial'] is not None and len(kind['trial']['starts']) == 2
    orig = t._enqueue_step

    def boom(*a, **k):
        raise RuntimeError('injected')
    t._enqueue_step = boom
    with pytest.raises(RuntimeError, match='injected'):
        t.batch(x, y, train=True)
    t._enqueue_step = orig
    assert kind['trial'] is None and kind['mode'] is None and not t._exec.enabled
    for _ in range(3 * (2 * t.TRIAL_STEPS + 2)):
        t.batch(x, y, train=True)
    assert t.graph_decided()


def test_replay_after_the_buffers_it_baked_in_were_replaced(tmp_path):
    """A captured step holds the addresses of buffers it does not own: the trainer's workspace, both networks' transformed-weight
    pools, the flat gradients, Adam's moments.  Capture; then (1) a training step of a larger extent (the workspace grows and is
    replaced), (2) set_tuning to other kernels and back (the weight pools are dropped and rebuilt), (3) a streamed inference pass
    through the generator on the default execution state -- and replay: the captured step keeps its own buffers alive, so the replay
    computes exactly what an eager trainer computes after the same sequence, bit for bit."""
    import patchgan_amd as pg
    from patchgan_amd import _lib as L
    from patchgan_amd.infer import predict_image
    gen = torch.Generator().manual_seed(6)
    x = torch.rand(2, 3, 256, 256, generator=gen)
    y = (torch.rand(2, 1, 256, 256, generator=gen) > 0.7).float()
    xb = torch.rand(6, 3, 256, 256, generator=gen)
    yb = (torch.rand(6, 1, 256, 256, generator=gen) > 0.7).float()
    img = torch.rand(3, 512, 512, generator=gen).cuda()
    outs = []
    for graph in (True, False):
        rows, gw, dw, used, t = _run(tmp_path, graph, 'fp32', 5, nf=32, tag=f'rep{int(graph)}')
        assert used[-1] == bool(graph)
        G, D = t.generator, t.discriminator
        ws_before = [b.data_ptr() for b in t._exec.buffers()]
        seq = []
        seq.append(t.batch(xb, yb, train=True))                       # (1) larger extent: eager, the workspace may grow ...
        t._exec.release()                                              # ... and is certainly replaced after this
        G.set_tuning(L.TUNE_WINO_OFF), D.set_tuning(L.TUNE_WINO_OFF)   # (2) other kernels (another kind of step) ...
        seq.append(t.batch(x, y, train=True))
        G.set_tuning(0), D.set_tuning(0)                               # ... and back: the captured kind's key matches again
        G.eval()
        m = predict_image(G, img, 256, 0.9, 0.0)                        # (3) inference on the default execution state
        G.train()
        seq.append(t.batch(x, y, train=True))                          # replay (graph=True) / eager
        seq.append(t.batch(x, y, train=True))
        if graph:
            assert t.launch_mode == 'graph'
            assert [b.data_ptr() for b in t._exec.buffers()] != ws_before          # the workspace really was replaced after the capture
        torch.cuda.synchronize()
        outs.append(([[float(l[k]) for k in LOSS_KEYS] for l in seq], G.flat.cpu().numpy().copy(), D.flat.cpu().numpy().copy(), m))
    assert outs[0][0] == outs[1][0], (outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2]) and np.array_equal(outs[0][3], outs[1][3])


def test_two_trainers_do_not_share_execution_state(tmp_path):
    """engine.Exec is per trainer (flags, held operands, workspaces; the second stream itself is the device's): two trainers
    alternating steps (one on two streams, one on one) give what each gives alone, and releasing one leaves the other's buffers alone."""
    from patchgan_amd import engine as E
    alone = _run(tmp_path, False, 'fp32', 4, nf=16, tag='al', two_streams=True)
    import patchgan_amd as pg
    ts = []
    for i, two in enumerate((True, False)):
        torch.manual_seed(99)
        g = pg.UNet(3, 1, 16, use_dropout=False, activation='leakyrelu', final_act='sigmoid').cuda()
        d = pg.Discriminator(4, 16, n_layers=3).cuda()
        t = pg.Trainer(g, d, str(tmp_path / f'pair{i}'))
        t.two_streams = two
        t.setup_optimizers(1e-3, 2e-3)
        g.train(), d.train()
        ts.append(t)
    gen = torch.Generator().manual_seed(5)
    rows = [[], []]
    for s in range(4):
        x = torch.rand(2, 3, 256, 256, generator=gen)
        y = (torch.rand(2, 1, 256, 256, generator=gen) > 0.7).float()
        for i, t in enumerate(ts):
            l = t.batch(x, y, train=True)
            rows[i].append([float(l[k]) for k in LOSS_KEYS])
    torch.cuda.synchronize()
    assert ts[0]._exec is not ts[1]._exec and ts[0]._exec.stream is not None and ts[1]._exec.stream is None
    assert np.array_equal(np.array(rows[0]), alone[0]) and np.array_equal(np.array(rows[1]), alone[0])
    ptrs = [b.data_ptr() for b in ts[1]._exec.buffers()]
    ts[0].release()
    assert ts[0]._exec.buffers() == [] and [b.data_ptr() for b in ts[1]._exec.buffers()] == ptrs
    E.release_workspaces()


def test_discriminator_update_in_flight_is_completed_by_whatever_reads_its_weights(tmp_path):
    """In a two-stream step the discriminator's backward pass and Adam update are still running on the second stream when batch()
    returns (they run under the next step's generator forward).  Reading the discriminator through its own surface -- state_dict(),
    forward(), save() -- completes the update first (the modules' access hook -> Trainer.flush), with NO device synchronize by the
    caller: the values are those of the one-stream trainer, bit for bit."""
    import patchgan_amd as pg
    outs = []
    for two in (True, False):
        torch.manual_seed(21)
        g = pg.UNet(3, 1, 32, use_dropout=False, activation='leakyrelu', final_act='sigmoid').cuda()
        d = pg.Discriminator(4, 32, n_layers=3).cuda()
        t = pg.Trainer(g, d, str(tmp_path / f'h{int(two)}'))
        t.two_streams = two
        t.setup_optimizers(1e-3, 1e-3)
        g.train(), d.train()
        gen = torch.Generator().manual_seed(5)
        x = torch.rand(4, 3, 256, 256, generator=gen)
        y = (torch.rand(4, 1, 256, 256, generator=gen) > 0.7).float()
        for _ in range(3):
            t.batch(x, y, train=True)
        if two:
            assert t._deferred is not None
        sd = {k: v.clone() for k, v in d.state_dict().items()}          # (no synchronize: the hook joins the streams)
        assert t._deferred is None
        t.batch(x, y, train=True)
        with torch.no_grad():
            out = d(torch.cat((x, y), 1).cuda())                         # the module's own forward: the 4th update is in place
        t.batch(x, y, train=True)
        t.save(1)                                                         # save() flushes too
        ck = torch.load(str(tmp_path / f'h{int(two)}' / 'discriminator_ep_001.pth'))
        outs.append((sd, out.cpu(), ck))
    for k in outs[0][0]:
        assert torch.equal(outs[0][0][k].cpu(), outs[1][0][k].cpu()), k
        assert torch.equal(outs[0][2][k].cpu(), outs[1][2][k].cpu()), k
    assert torch.equal(outs[0][1], outs[1][1])


def test_bounded_operand_holding_joins_early_and_changes_nothing(tmp_path, monkeypatch):
    """engine.KEEP_LIMIT_BYTES bounds what the second stream keeps referenced inside a backward pass: past it the streams join in the
    middle of the pass (and the held operands are dropped).  With the limit at 1 MiB every layer joins: same losses and weights as the
    one-stream step, bit for bit -- and the early joins really happened."""
    from patchgan_amd import engine as E
    ref = _run(tmp_path, False, 'fp32', 5, nf=32, tag='k_ref')
    monkeypatch.setattr(E, 'KEEP_LIMIT_BYTES', 1 << 20)
    peaks = []
    orig = E.ConvOp.wgrad

    def spy(self, *a, **k):
        r = orig(self, *a, **k)
        peaks.append(E.cur_exec().keep_bytes)
        return r
    monkeypatch.setattr(E.ConvOp, 'wgrad', spy)
    two = _run(tmp_path, False, 'fp32', 5, nf=32, tag='k_two', two_streams=True)
    assert peaks and max(peaks) <= (1 << 20) and peaks.count(0) > len(peaks) // 2, peaks[:20]
    assert np.array_equal(ref[0], two[0]) and np.array_equal(ref[1], two[1]) and np.array_equal(ref[2], two[2])


def test_two_stream_step_that_runs_out_of_memory_falls_back_to_one_stream(tmp_path, monkeypatch):
    """The two-stream step holds about twice the one-stream step's device memory.  A torch.cuda.OutOfMemoryError raised inside a two-stream
    step before anything is committed (here: injected at the first weight gradient of the SECOND step's backward pass, i.e. with the first
    step's discriminator update still running on the second stream and this step's early discriminator forward forked): the trainer joins
    and drops the second stream's state, warns once, runs THAT step again on one stream and pins the kind to one stream -- same losses and
    weights as the one-stream trainer.  (The allocator frees its cache and retries before it raises, so a memory cap between the two
    peaks does not provoke the error reliably: the second half of the test runs under such a cap and accepts either outcome.)"""
    import warnings
    from patchgan_amd import engine as E
    E.release_workspaces()
    torch.cuda.empty_cache()

    def peak(two, tag, steps=3):
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        r = _run(tmp_path, False, 'fp32', steps, nf=32, tag=tag, two_streams=two)
        m = torch.cuda.max_memory_allocated()
        r[4].release()
        return r, m
    one, m1 = peak(False, 'oom1')
    two, m2 = peak(True, 'oom2')
    assert np.array_equal(one[0], two[0])
    assert m2 > m1 * 1.2, (m1, m2)                 # (the premise: two streams need visibly more)
    del two
    import patchgan_amd as pg
    real_wgrad, real_enqueue = E.ConvOp.wgrad, pg.Trainer._enqueue_step
    seen = {'steps': 0, 'raised': 0}

    def counting(self, *a, **k):
        seen['steps'] += 1
        return real_enqueue(self, *a, **k)

    def failing(self, *a, **k):
        if E.cur_exec().enabled and seen['steps'] == 2 and not seen['raised']:       # the first weight gradient of step 2's backward pass
            seen['raised'] = 1
            raise torch.cuda.OutOfMemoryError('injected: HIP out of memory')
        return real_wgrad(self, *a, **k)
    monkeypatch.setattr(pg.Trainer, '_enqueue_step', counting)
    monkeypatch.setattr(E.ConvOp, 'wgrad', failing)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        inj = _run(tmp_path, False, 'fp32', 3, nf=32, tag='oom_inj', two_streams=True)
    monkeypatch.setattr(E.ConvOp, 'wgrad', real_wgrad)
    monkeypatch.setattr(pg.Trainer, '_enqueue_step', real_enqueue)
    t = inj[4]
    assert seen['raised'] == 1 and seen['steps'] == 4
    assert t.oom_fallbacks == 1 and len(t._oom_kinds) == 1, (t.oom_fallbacks, seen)
    assert any('out of device memory' in str(x.message) for x in w), [str(x.message) for x in w]
    assert t.launch_mode == 'eager1'
    assert np.array_equal(one[0], inj[0]) and np.array_equal(one[1], inj[1]) and np.array_equal(one[2], inj[2])
    t.redecide()
    assert not t._oom_kinds
    t.release()
    del inj, t
    # the real thing, where the allocator lets it happen: a cap between the two peaks
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    base = torch.cuda.memory_reserved()
    total = torch.cuda.get_device_properties(0).total_memory
    cap = base + m1 + (m2 - m1) // 4
    torch.cuda.set_per_process_memory_fraction(min(1.0, cap / total))
    try:
        try:
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter('always')
                capped = _run(tmp_path, False, 'fp32', 3, nf=32, tag='oom3', two_streams=True)
        except torch.cuda.OutOfMemoryError:
            capped = None          # (fragmentation: not even the one-stream step fitted under this cap -- nothing to compare)
        if capped is not None:
            t = capped[4]
            assert t.oom_fallbacks in (0, 1)
            if t.oom_fallbacks:
                assert t.launch_mode == 'eager1' and any('out of device memory' in str(x.message) for x in w)
            assert np.array_equal(one[0], capped[0]) and np.array_equal(one[1], capped[1]) and np.array_equal(one[2], capped[2])
    finally:
        torch.cuda.set_per_process_memory_fraction(1.0)
        E.release_workspaces()
        torch.cuda.empty_cache()


def test_captured_two_stream_step_is_bit_identical(tmp_path, monkeypatch):
    """'graph2' (by decree only): the fork / join schedule of the two-stream step inside the captured hipGraph -- the second stream joins
    the capture through the engine's own stream waits, every chain is joined before the step ends, the discriminator's backward pass is
    not deferred across the end of a captured step.  Same losses and weights as launch by launch on one stream."""
    import patchgan_amd as pg
    ref = _run(tmp_path, False, 'fp32', 8, nf=16, tag='g2_ref')
    monkeypatch.setattr(pg.Trainer, 'AUTO_FORCE', 'graph2')
    g2 = _run(tmp_path, 'auto', 'fp32', 8, nf=16, tag='g2')
    assert g2[4].launch_mode == 'graph2' and g2[3][-1], (g2[4].launch_mode, g2[3])
    assert np.array_equal(ref[0], g2[0]) and np.array_equal(ref[1], g2[1]) and np.array_equal(ref[2], g2[2])


def test_tournament_periods_do_not_span_steps_of_another_kind(tmp_path):
    """'auto' scores a candidate by start-to-start periods of consecutive steps of ONE kind.  With evaluation passes between the
    training steps no period of the training kind is clean: its trial keeps starting over (no decision from periods that contain another
    kind's step), the steps run on the candidate under trial, and the results stay those of the one-stream trainer."""
    ref = _run(tmp_path, False, 'fp32', 8, nf=16, tag='ik_ref', eval_at=range(8))
    auto = _run(tmp_path, 'auto', 'fp32', 8, nf=16, tag='ik_auto', eval_at=range(8))
    t = auto[4]
    assert not t.graph_decided()
    for kind in t._kinds.values():
        assert kind['mode'] is None and (kind['trial'] is None or len(kind['trial']['starts']) <= 1), kind
    assert np.array_equal(ref[0], auto[0]) and np.array_equal(ref[1], auto[1]) and np.array_equal(ref[2], auto[2])


def test_optimizer_reset_and_a_second_trainer_wait_for_the_update_in_flight(tmp_path):
    """setup_optimizers() (every train() call re-creates Adam, trainer.py:169-172) and a second Trainer built on the same networks both
    touch state that a discriminator update still running on the second stream reads and writes: both complete it first.  Same
    weights as the one-stream trainer doing the same thing."""
    import patchgan_amd as pg
    outs = []
    for two in (True, False):
        torch.manual_seed(21)
        g = pg.UNet(3, 1, 16, use_dropout=False, activation='leakyrelu', final_act='sigmoid').cuda()
        d = pg.Discriminator(4, 16, n_layers=3).cuda()
        t = pg.Trainer(g, d, str(tmp_path / f'o{int(two)}'))
        t.two_streams = two
        t.setup_optimizers(1e-3, 1e-3)
        g.train(), d.train()
        gen = torch.Generator().manual_seed(5)
        x = torch.rand(4, 3, 256, 256, generator=gen)
        y = (torch.rand(4, 1, 256, 256, generator=gen) > 0.7).float()
        for _ in range(3):
            t.batch(x, y, train=True)
        if two:
            assert t._deferred is not None
        t.setup_optimizers(5e-4, 5e-4)                 # fresh moments while Adam(D) of step 3 may still be running
        assert t._deferred is None
        for _ in range(2):
            t.batch(x, y, train=True)
        if two:
            assert t._deferred is not None
        t2 = pg.Trainer(g, d, str(tmp_path / f'o{int(two)}b'))      # a second driver of the same networks
        assert t._deferred is None
        t2.two_streams = two
        t2.setup_optimizers(1e-3, 1e-3)
        t2.batch(x, y, train=True)
        t2.flush()
        torch.cuda.synchronize()
        outs.append((g.flat.cpu().clone(), d.flat.cpu().clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
