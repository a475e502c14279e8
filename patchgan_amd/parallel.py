"""Data parallelism for the G+D step: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
MI355X; "gloo" in the CPU tests).  The reference has no distributed code at all: this is new.

What is exchanged per step:
  * SUM all-reduce of the flat generator gradient buffer (167 MB at nf=64) and of the flat discriminator gradient
    buffer (11 MB).  `GradReducer` cuts the flat buffer into buckets in the order backward produces them (the last
    layer's block first) and hands each bucket to the collective as soon as its last weight gradient has been
    enqueued.  On a HIP device the collectives are issued from a SECOND HIP stream (`Dist.comm_stream`) that waits on
    an event recorded behind the bucket's last kernel, so they run under the remaining backward kernels and under the
    discriminator step that follows; the optimizer's stream waits on the comm stream.  xGMI is point to point
    (7 links x ~153 GB/s): buckets are kept large (default 32 MiB) because a ring all-reduce is per-link bound, not
    latency bound.
  * two fp64 scalars per generator loss (sum_b (1 - T_b) and sum(y)): focal-Tversky's batch mean under the power
    and weighted-BCE's global sum(y) are non-linear in the batch, so each rank needs the GLOBAL value before it
    can seed its local gradient (engine.loss_value_and_grad); issued on the comm stream too, under the
    discriminator's forward pass over the fake batch.
  * the step's loss scalars for logging.
InstanceNorm statistics are per sample: nothing else crosses ranks.
"""
import os

import torch


class Dist:
    """Thin view of torch.distributed; an uninitialised / single-rank group degrades to no-ops.
    PATCHGAN_DP_FORCE=1 keeps the data-parallel code path on for a one-rank group (exercises the RCCL calls and the
    comm-stream ordering on a single-GPU box)."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        ready = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if ready else 1
        self.rank = dist.get_rank() if ready else 0
        self.on = ready and (self.world > 1 or os.environ.get('PATCHGAN_DP_FORCE') == '1')
        self.backend = dist.get_backend() if ready else None
        self._comm = None
        self.timing = None          # list of (start_event, end_event, nbytes) when bench.py asks for it
        self.exposed = None         # list of (before_wait, after_wait, nbytes): see all_reduce_side

    def all_reduce(self, t, async_op=False):
        if self.on:
            return self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, async_op=async_op)
        return None

    # ---- second-stream collectives (device tensors) ------------------------------------------------------------
    def comm_stream(self, device):
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=device)
        return self._comm

    def all_reduce_side(self, t, producers=()):
        """SUM all-reduce of `t` ordered after everything enqueued so far on the current stream (and on every stream in
        `producers`: a two-stream step writes weight gradients on its second stream), without holding that stream up.
        Returns wait(): call it (on the stream that consumes `t`) before the result is read.  Device tensors go through
        the comm stream; host tensors (gloo CPU tests) through an async work handle."""
        if not self.on:
            return lambda: None
        if not t.is_cuda:
            h = self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, async_op=True)
            return _HostWait(h)
        comm = self.comm_stream(t.device)
        ready = torch.cuda.Event()
        ready.record()                                   # behind the producer kernels on the compute stream
        with torch.cuda.stream(comm):
            comm.wait_event(ready)
            for p in producers:
                comm.wait_stream(p)
            if self.timing is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            # a blocking-mode collective: ProcessGroupNCCL enqueues it on its own stream behind `comm` and makes `comm`
            # wait for it, so `comm` is complete exactly when the reduced values are in place
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            if self.timing is not None:
                e1.record()
                self.timing.append((e0, e1, t.numel() * t.element_size()))
        done = torch.cuda.Event()
        done.record(comm)
        if self.timing is None:
            return _StreamWait(done, comm)
        nbytes = t.numel() * t.element_size()

        def wait_timed():
            # EXPOSED communication: how long the consuming stream really stood still for this collective = the time between an
            # event recorded just before the wait and one recorded just after it (~0 when the collective had already finished
            # under the kernels that were enqueued in between)
            cur = torch.cuda.current_stream()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(cur)
            cur.wait_event(done)
            b.record(cur)
            if self.exposed is not None:
                self.exposed.append((a, b, nbytes))
        wait_timed.in_order = False          # a timed wait is measured one by one: never folded into a later one
        return wait_timed


class _HostWait:
    """wait() of a host-tensor collective (gloo): every work handle has to be waited for."""
    in_order = False

    def __init__(self, work):
        self.work = work

    def __call__(self):
        self.work.wait()


class _StreamWait:
    """wait() of a device collective: the consuming stream waits for the event recorded on the comm stream behind it.  All device
    collectives of a process go through ONE comm stream, in issue order, so waiting for a later one covers every earlier one
    (`in_order`): GradReducer.finish() puts one wait on the compute stream instead of one per bucket (a stream wait or an event
    record is a barrier packet, 5-10 us of an otherwise back-to-back kernel queue on MI355X, tools/dp_sync_probe.py)."""
    in_order = True

    def __init__(self, done, comm=None):
        self.done, self.comm = done, comm

    def __call__(self):
        torch.cuda.current_stream().wait_event(self.done)


_CURRENT = None


def current():
    """The process's Dist, rebuilt only when torch.distributed's state changed (Trainer.batch calls this every step)."""
    global _CURRENT
    import torch.distributed as dist
    ready = dist.is_available() and dist.is_initialized()
    if _CURRENT is None or (_CURRENT.backend is not None) != ready:
        _CURRENT = Dist()
    return _CURRENT


class GradReducer:
    """Bucketed, asynchronous SUM all-reduce of a flat gradient buffer that is filled from its END towards its START
    (backward visits layers last to first, and the flat buffer is laid out first layer to last)."""

    def __init__(self, dist, flat, bucket_bytes=32 << 20, producers=None):
        """producers(): streams besides the current one that may hold a ready range's last writes (engine.side_producers)."""
        self.dist, self.flat, self.producers = dist, flat, producers
        self.bucket_elems = max(1, bucket_bytes // flat.element_size())
        self.hi = flat.numel()      # everything in [hi, numel) has been handed to a collective
        self.lo = flat.numel()      # everything in [lo, hi) is ready but not yet launched
        self.waits = []
        self.launched = []          # (lo, hi) ranges, for tests

    def ready(self, lo, hi):
        """Gradients of flat[lo:hi] are complete (enqueued on the current stream).  Ranges must arrive contiguous and
        descending: hi == the previous lo."""
        if hi != self.lo:
            raise RuntimeError(f"GradReducer: expected a range ending at {self.lo}, got [{lo}, {hi})")
        self.lo = lo
        if self.hi - self.lo >= self.bucket_elems:
            self._launch()

    def _launch(self):
        if self.lo < self.hi:
            extra = self.producers() if (self.producers is not None and self.flat.is_cuda) else ()
            self.waits.append(self.dist.all_reduce_side(self.flat[self.lo:self.hi], extra) if extra else
                              self.dist.all_reduce_side(self.flat[self.lo:self.hi]))
            self.launched.append((self.lo, self.hi))
            self.hi = self.lo

    def finish(self, launch_only=False):
        """Launch whatever is left (down to element 0) and, unless `launch_only`, make the current stream wait for every bucket."""
        self.lo = 0
        self._launch()
        if launch_only:
            return
        last = self.waits[-1] if self.waits else None
        for i, w in enumerate(self.waits):
            # covered by the wait for the last bucket ONLY if both are event waits behind the same in-order comm stream (a process group
            # that hands back anything else -- a host-side work handle, another stream -- gets every wait)
            if (i + 1 < len(self.waits) and isinstance(w, _StreamWait) and isinstance(last, _StreamWait)
                    and w.comm is not None and w.comm is last.comm):
                continue
            w()
        self.waits = []


def shard_batch(x, y, rank, world):
    """Contiguous B/world shard of a global batch (global batch split contiguously per rank)."""
    B = x.shape[0]
    if B % world:
        raise ValueError(f"global batch {B} is not divisible by world size {world}")
    per = B // world
    return x[rank * per:(rank + 1) * per], y[rank * per:(rank + 1) * per]
