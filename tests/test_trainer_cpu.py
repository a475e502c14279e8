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


def test_execution_state_is_per_owner_and_per_thread():
    """engine.Exec (workspaces, second stream, book-keeping of a two-stream step) belongs to ONE driver: `with exec:` makes it current on
    the calling thread only, nests (Trainer.flush() inside Trainer.batch() enters the same state again) and restores what was current
    before; another thread sees its own state or none.  No GPU needed: nothing here allocates."""
    import threading
    from patchgan_amd import engine as E
    a, b = E.Exec(), E.Exec()
    assert getattr(E._TLS, 'cur', None) is None
    seen = {}
    with a:
        assert E._TLS.cur is a
        with a:                                   # re-entered by the same owner
            with b:
                assert E._TLS.cur is b
            assert E._TLS.cur is a
        assert E._TLS.cur is a

        def other():
            seen['before'] = getattr(E._TLS, 'cur', None)
            with b:
                seen['inside'] = E._TLS.cur
            seen['after'] = getattr(E._TLS, 'cur', None)
        th = threading.Thread(target=other)
        th.start()
        th.join()
        assert E._TLS.cur is a                    # the other thread's `with b:` did not touch this thread's current state
    assert getattr(E._TLS, 'cur', None) is None
    assert seen == {'before': None, 'inside': b, 'after': None}
    assert not a.keep and not a.pending and a.buffers() == [] and a.stream is None
    a.release()                                    # nothing to join, nothing to free: no GPU call
    assert a.ws_gen == 1


def test_launch_mode_policy_without_a_gpu():
    """Trainer._launch_mode is host logic: what `graph` / `two_streams` allow, the warm-up count, forced modes.  (The timed 'auto'
    decision needs device events: tests/test_graph_gpu.py.)"""
    import types
    from patchgan_amd.trainer import Trainer
    t = Trainer.__new__(Trainer)
    t._kinds, t._graphs, t.step_times, t._oom_kinds = {}, {}, None, set()
    t.generator = types.SimpleNamespace(training=True, engine=types.SimpleNamespace(use_dropout=False))
    t.graph, t.two_streams = False, None
    assert [t._launch_mode('k', True) for _ in range(5)] == ['eager1'] * 5 and t._kinds == {}
    # a kind that ran out of device memory on two streams stays on one stream whatever the setting says
    t.two_streams, t._oom_kinds = True, {'k'}
    assert t._launch_mode('k', True) == 'eager1' and t._launch_mode('other', True) == 'eager2'
    t.two_streams, t._oom_kinds = None, set()
    t.two_streams = True
    assert t._launch_mode('k', True) == 'eager2' and t._launch_mode('e', False) == 'eager2'      # forced: no warm-up, eval passes too
    t.graph, t.two_streams = True, None
    assert [t._launch_mode('k', True) for _ in range(5)] == ['eager1'] * 3 + ['graph'] * 2        # captured at the 4th step of its kind
    assert t._launch_mode('e', False) == 'eager1'                                                # never an evaluation pass
    t.generator.engine.use_dropout = True                                                        # dropout: launch arguments change per step
    assert t._launch_mode('k', True) == 'eager1' and t._launch_mode('d', True) == 'eager1'
    t.generator.engine.use_dropout = False
    t.graph, t._kinds, t.AUTO_FORCE = 'auto', {}, 'eager2'                                       # 'auto' by decree: after the warm steps
    assert [t._launch_mode('k', True) for _ in range(5)] == ['eager1'] * 3 + ['eager2'] * 2 and t.decided_modes() == ['eager2']
    t._kinds, t.AUTO_FORCE = {}, 'graph'
    assert [t._launch_mode('k', True) for _ in range(4)] == ['eager1'] * 3 + ['graph']
    assert [t._launch_mode('e', False) for _ in range(4)] == ['eager1'] * 4                       # an evaluation pass cannot be captured
    assert t.decided_modes() == ['graph', 'eager1'] and t.graph_decided()
