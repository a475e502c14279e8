"""world_size-2 gloo tests (CPU) of the data-parallel plumbing: bucketed asynchronous gradient all-reduce, batch
sharding, and the exactness recipe for the batch-non-linear losses, checked with the CPU oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)


def _reducer_worker(rank, world, port, q):
    _init(rank, world, port)
    from patchgan_amd.parallel import Dist, GradReducer
    d = Dist()
    assert d.on and d.world == world and d.rank == rank
    n = 1000
    flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
    r = GradReducer(d, flat, bucket_bytes=4 * 300)          # 300-element buckets
    # layers complete last-to-first in uneven pieces
    edges = [1000, 900, 650, 640, 300, 120, 0]
    for hi, lo in zip(edges[:-1], edges[1:]):
        r.ready(lo, hi)
    with pytest.raises(RuntimeError):
        r.ready(500, 600)                                    # not contiguous with what came before
    r.finish()
    want = torch.arange(n, dtype=torch.float32) * sum(range(1, world + 1))
    ok = torch.equal(flat, want)
    # buckets: launched when >= 300 elements are pending, remainder at finish; together they tile [0, n)
    ranges = sorted(r.launched)
    tiles = ranges[0][0] == 0 and ranges[-1][1] == n and all(a[1] == b[0] for a, b in zip(ranges[:-1], ranges[1:]))
    q.put((rank, ok, tiles, r.launched))
    dist.destroy_process_group()


def test_grad_reducer_waits_once_for_in_order_collectives():
    """Device collectives share one in-order comm stream: finish() waits for the last bucket only (one barrier packet on the
    compute stream instead of one per bucket) -- but ONLY for event waits that sit behind the SAME comm stream (_StreamWait with that
    stream); host work handles, timed waits, waits of another stream or of an unknown kind are waited for one by one;
    finish(launch_only) hands the remainder over without waiting."""
    from patchgan_amd.parallel import GradReducer, _StreamWait

    class Waited(_StreamWait):
        def __init__(self, log, i, comm):
            super().__init__(None, comm)
            self.log, self.i = log, i

        def __call__(self):
            self.log.append(self.i)

    class FakeDist:
        def __init__(self, kind):
            self.calls, self.waited, self.kind = [], [], kind
            self.comm = object()

        def all_reduce_side(self, t):
            i = len(self.calls)
            self.calls.append(t.numel())
            if self.kind == 'one_stream':
                return Waited(self.waited, i, self.comm)
            if self.kind == 'stream_per_call':
                return Waited(self.waited, i, object())
            if self.kind == 'no_stream':
                return Waited(self.waited, i, None)

            def wait():                      # a host-side handle that merely CLAIMS to be in order
                self.waited.append(i)
            wait.in_order = True
            return wait

    for kind, want in (('one_stream', [3]), ('stream_per_call', [0, 1, 2, 3]), ('no_stream', [0, 1, 2, 3]), ('callable', [0, 1, 2, 3])):
        d = FakeDist(kind)
        r = GradReducer(d, torch.zeros(1000), bucket_bytes=4 * 300)
        for hi, lo in ((1000, 700), (700, 400), (400, 100)):
            r.ready(lo, hi)
        assert d.calls == [300, 300, 300] and d.waited == []
        r.finish(launch_only=True)
        assert d.calls == [300, 300, 300, 100] and d.waited == []
        r.finish()
        assert d.calls == [300, 300, 300, 100] and d.waited == want, (kind, d.waited)
        assert sorted(r.launched) == [(0, 100), (100, 400), (400, 700), (700, 1000)]


def test_grad_reducer_hands_the_collective_every_producer_stream():
    """Two-stream step under data parallelism: a bucket's weight gradients may have been written on the second stream, so the reducer
    passes `producers()` on to all_reduce_side for device tensors (the comm stream then waits for those streams too) -- and calls the
    plain one-argument form when there is nothing to add or the tensor lives on the host."""
    from patchgan_amd.parallel import GradReducer

    class Flat:
        """a stand-in with the three things GradReducer touches"""
        def __init__(self, n, cuda):
            self.n, self.is_cuda = n, cuda

        def numel(self):
            return self.n

        def element_size(self):
            return 4

        def __getitem__(self, sl):
            return (sl.start, sl.stop)

    class FakeDist:
        def __init__(self):
            self.calls = []

        def all_reduce_side(self, t, producers=None):
            self.calls.append((t, producers))
            return lambda: None

    side = object()
    for cuda, prod, want in ((True, lambda: [side], [side]), (True, lambda: [], None), (False, lambda: [side], None), (True, None, None)):
        d = FakeDist()
        r = GradReducer(d, Flat(1000, cuda), bucket_bytes=4 * 500, producers=prod)
        r.ready(500, 1000)
        r.ready(0, 500)
        r.finish()
        assert d.calls == [((500, 1000), want), ((0, 500), want)], (cuda, d.calls)


def test_grad_reducer_world2():
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, tiles, launched in res:
        assert ok and tiles, (rank, launched)
        assert launched[0] == (650, 1000) and launched[1] == (300, 650), launched


def _exactness_worker(rank, world, port, q, loss_type):
    """Each rank differentiates ITS shard with the loss seeded by the global terms; the SUM of the per-rank gradients
    must equal the single-process gradient of the whole batch (what Trainer + pg_loss_finalize implement on the GPU)."""
    _init(rank, world, port)
    from oracle import patchgan_oracle as O
    from patchgan_amd.parallel import Dist, shard_batch
    d = Dist()
    g = torch.Generator().manual_seed(11)
    B, C = 4, 3
    p_all = torch.rand(B, C, 16, 16, generator=g).clamp(0.01, 0.99)
    y_all = (torch.rand(B, C, 16, 16, generator=g) > 0.6).float()
    # single-process reference
    pr = p_all.clone().requires_grad_(True)
    full = O.seg_loss(loss_type, pr, y_all, 200)
    full.backward()
    p, y = shard_batch(p_all, y_all, rank, world)
    p = p.clone().requires_grad_(True)
    if loss_type == 'tversky':
        tp = (y * p).sum((1, 2, 3)); fn = ((1 - p) * y).sum((1, 2, 3)); fp = (p * (1 - y)).sum((1, 2, 3))
        one_minus_t = 1 - (tp + 1) / (tp + 0.75 * fn + 0.25 * fp + 1)
        s = one_minus_t.sum().detach().clone()
        d.all_reduce(s)                                        # global sum_b (1 - T_b)
        m = s / B
        # d/dp [200 m^g] = 200 g m^(g-1) / B * d(sum_b (1-T_b))/dp : seed the LOCAL sum with the GLOBAL factor
        local = (200 * 0.75 * m ** (0.75 - 1) / B) * one_minus_t.sum()
        value = 200 * m ** 0.75
    elif loss_type == 'weighted_bce':
        sy = y.sum().clone()
        d.all_reduce(sy)                                       # global sum(y)
        w = 1 - y.sum((2, 3), keepdim=True) / sy
        local = 200 * (torch.nn.functional.binary_cross_entropy(p, y, weight=w, reduction='sum') / (B * C * 256))
        value = local.detach().clone()
        d.all_reduce(value)
    else:
        local = 200 * (p - y).abs().sum() / (B * C * 256)
        value = local.detach().clone()
        d.all_reduce(value)
    local.backward()
    gfull = torch.zeros_like(p_all)
    per = B // world
    gfull[rank * per:(rank + 1) * per] = p.grad
    d.all_reduce(gfull)
    err = ((gfull - pr.grad).abs().max() / pr.grad.abs().max()).item()
    q.put((rank, err, abs(float(value) - full.item()) / abs(full.item())))
    dist.destroy_process_group()


@pytest.mark.parametrize('loss_type', ['tversky', 'weighted_bce', 'MAE'])
def test_sharded_loss_seeds_are_exact(loss_type):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_exactness_worker, args=(r, 2, port, q, loss_type)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, gerr, verr in res:
        assert gerr < 1e-5 and verr < 1e-5, (rank, gerr, verr)


def test_shard_batch():
    from patchgan_amd.parallel import shard_batch
    x = torch.arange(8).view(8, 1)
    a, b = shard_batch(x, x, 1, 4)
    assert a.flatten().tolist() == [2, 3]
    with pytest.raises(ValueError):
        shard_batch(x, x, 0, 3)


def test_train_reseeds_distributed_sampler_each_epoch(tmp_path):
    """Under data parallelism the loaders carry a DistributedSampler: Trainer.train must call set_epoch(epoch) or every epoch
    repeats the first permutation (and the same per-rank shards).  Host-only: batch() is stubbed out."""
    import patchgan_amd as pg
    from torch.utils.data import DataLoader, TensorDataset
    from torch.utils.data.distributed import DistributedSampler
    ds = TensorDataset(torch.arange(32).float().view(32, 1), torch.arange(32).float().view(32, 1))
    loader = DataLoader(ds, batch_size=4, sampler=DistributedSampler(ds, num_replicas=2, rank=0, shuffle=True, seed=3))
    val = DataLoader(ds, batch_size=8, sampler=DistributedSampler(ds, num_replicas=2, rank=0, shuffle=False))

    class Recorder(pg.Trainer):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.seen = []

        def setup_optimizers(self, *a, **k):
            pass

        def batch(self, x, y, train=False):
            if train:
                self.seen.append(x.flatten().tolist())
            return dict(gen=1.0, gen_loss=1.0, gdisc=0.0, discr=0.5, discf=0.5, disc=0.5)

    t = Recorder(pg.UNet(3, 1, 4), pg.Discriminator(4, 4), str(tmp_path))
    G_ep, D_ep = t.train(loader, val, 2, save_freq=100)
    assert G_ep == [1.0, 1.0] and D_ep == [0.5, 0.5]
    per_epoch = len(loader)
    first, second = t.seen[:per_epoch], t.seen[per_epoch:]
    assert len(second) == per_epoch and first != second
    assert sorted(sum(first, [])) != list(range(32))          # this rank sees only its half
    assert len(set(sum(first, []))) == 16


# ---- the multi-rank launcher of bench.py (`python bench.py --gpus N`): a dead rank must end the job, not hang it ---------------


def _spawn(mode, *extra, timeout=60.0):
    """bench.spawn_ranks over tests/dp_kill_worker.py (world 2, gloo, CPU).  Returns (SystemExit message or None, seconds)."""
    import time
    import bench
    here = os.path.dirname(os.path.abspath(__file__))
    t0 = time.monotonic()
    try:
        bench.spawn_ranks(2, [mode, *map(str, extra)], script=os.path.join(here, 'dp_kill_worker.py'), timeout=timeout, grace=2.0)
        msg = None
    except SystemExit as e:
        msg = str(e.code)
    return msg, time.monotonic() - t0


def test_spawn_ranks_forwards_rank0_line(capfd):
    msg, _ = _spawn('ok')
    assert msg is None
    import json
    lines = [l for l in capfd.readouterr().out.splitlines() if l.startswith('{')]
    assert len(lines) == 1 and json.loads(lines[0]) == {'ok': True, 'value': 1.0}


@pytest.mark.parametrize('mode,die_rank,status', [('midstep', 1, 7), ('midstep', 0, 7), ('noshow', 1, 9)])
def test_spawn_ranks_ends_the_job_when_a_rank_dies(mode, die_rank, status):
    """A rank that dies mid-step leaves its peer inside a collective, one that never reaches the rendezvous leaves its peer in
    init_process_group (group timeout: 600 s).  The launcher must notice the first non-zero exit, terminate the survivor, name the
    failed rank and return non-zero -- within seconds, not after the collective timeout."""
    msg, dt = _spawn(mode, die_rank, timeout=120.0)
    assert msg is not None and f'rank {die_rank} exited with status {status}' in msg, msg
    assert dt < 60.0, dt


def test_spawn_ranks_overall_timeout():
    msg, dt = _spawn('hang', timeout=8.0)
    assert msg is not None and 'still running after 8 s' in msg, msg
    assert dt < 40.0, dt
