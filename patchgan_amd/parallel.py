"""Data parallelism for the G+D step: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
MI355X; "gloo" in the CPU tests).  The reference has no distributed code at all: this is new.

What is exchanged per step:
  * SUM all-reduce of the flat generator gradient buffer (167 MB at nf=64) and of the flat discriminator gradient
    buffer (11 MB).  `GradReducer` cuts the flat buffer into buckets in the order backward produces them (the last
    layer's block first) and launches each bucket's all-reduce asynchronously as soon as its last weight gradient has
    been enqueued, so the collective runs on RCCL's stream under the remaining backward kernels and under the
    discriminator step that follows; the optimizer waits on the handles.  xGMI is point to point (7 links x ~153 GB/s):
    buckets are kept large (default 32 MiB) because a ring all-reduce is per-link bound, not latency bound.
  * two fp64 scalars per generator loss (sum_b (1 - T_b) and sum(y)): focal-Tversky's batch mean under the power
    and weighted-BCE's global sum(y) are non-linear in the batch, so each rank needs the GLOBAL value before it
    can seed its local gradient (engine.loss_value_and_grad).
  * the step's loss scalars for logging.
InstanceNorm statistics are per sample: nothing else crosses ranks.
"""
import torch


class Dist:
    """Thin view of torch.distributed; an uninitialised / single-rank group degrades to no-ops."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.world = dist.get_world_size() if self.on else 1
        self.rank = dist.get_rank() if self.on else 0

    def all_reduce(self, t, async_op=False):
        if self.on:
            return self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, async_op=async_op)
        return None


class GradReducer:
    """Bucketed, asynchronous SUM all-reduce of a flat gradient buffer that is filled from its END towards its START
    (backward visits layers last to first, and the flat buffer is laid out first layer to last)."""

    def __init__(self, dist, flat, bucket_bytes=32 << 20):
        self.dist, self.flat = dist, flat
        self.bucket_elems = max(1, bucket_bytes // flat.element_size())
        self.hi = flat.numel()      # everything in [hi, numel) has been handed to a collective
        self.lo = flat.numel()      # everything in [lo, hi) is ready but not yet launched
        self.handles = []
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
            h = self.dist.all_reduce(self.flat[self.lo:self.hi], async_op=True)
            if h is not None:
                self.handles.append(h)
            self.launched.append((self.lo, self.hi))
            self.hi = self.lo

    def finish(self):
        """Launch whatever is left (down to element 0) and make the current stream wait for every bucket."""
        self.lo = 0
        self._launch()
        for h in self.handles:
            h.wait()
        self.handles = []


def shard_batch(x, y, rank, world):
    """Contiguous B/world shard of a global batch (global batch split contiguously per rank)."""
    B = x.shape[0]
    if B % world:
        raise ValueError(f"global batch {B} is not divisible by world size {world}")
    per = B // world
    return x[rank * per:(rank + 1) * per], y[rank * per:(rank + 1) * per]
