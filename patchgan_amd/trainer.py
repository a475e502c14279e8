"""Trainer with the reference's surface (patchgan/trainer.py:16-321): ``batch`` (one generator + one
discriminator update), ``train`` (epoch driver, Adam, LR schedule, checkpoints), ``save`` / ``load`` /
``load_last_checkpoint`` -- the G+D step hand-scheduled on the HIP engines instead of autograd.

Per-step schedule (reference trainer.py:50-115):
   x, y -> NHWC slices of ONE discriminator-input buffer din[2N] (real half = x|y, fake half = x|G(x)):
   G fwd (dec6 writes straight into the fake half) -> D fwd(fake) -> seg loss + BCE(D(fake),1) ->
   dgrad through D (its weight grads are skipped: the reference zeroes them at trainer.py:93-94) ->
   G bwd -> Adam(G) -> D fwd over din[2N] (real and detached fake in one batch) -> BCE halves -> D bwd -> Adam(D) ->
   one device->host copy of the loss scalars.
   Under data parallelism: G's gradient buckets are all-reduced asynchronously from inside G bwd and Adam(G) moves behind
   the D backward; D's gradient is all-reduced asynchronously and Adam(D) moves behind the next step's G forward (flush()).
   On two streams (a device-bound kind of step; engine.Exec) the same launches are spread over two in-order chains: the weight
   gradients of each backward pass, the D forward over din[2N] (enqueued as soon as G's output exists) and the whole D backward pass +
   Adam(D) (which then run under the NEXT step's G forward) go to the second stream; flush() joins before D's weights are read.
   How a kind of step is launched -- one stream, two streams, a replay of the captured hipGraph -- is measured, not guessed
   (_launch_mode: a tournament over the candidates the settings allow).

Data parallelism: one process per GPU (torch.distributed, backend "nccl" = RCCL).  InstanceNorm is per sample,
so the only exchanges are the SUM all-reduce of the flat gradient buffers and of two loss-normalisation terms
(focal-Tversky's batch mean and weighted-BCE's sum(y) are non-linear in the batch).  Gradient seeds are scaled
by the GLOBAL batch so the summed gradient equals the single-process large-batch gradient.
"""
import glob
import os
import time
from collections import defaultdict

import numpy as np
import torch
import tqdm

from . import _lib as L
from . import engine as E
from .parallel import current as _dist, GradReducer

device = 'cuda' if torch.cuda.is_available() else 'cpu'

_LOSS_MODES = {'tversky': L.LOSS_TVERSKY, 'weighted_bce': L.LOSS_WBCE, 'MAE': L.LOSS_MAE}
EARLY_D_FWD = E._exp_env('PATCHGAN_EARLY_D_FWD') != '0'      # two-stream step: the discriminator step's forward under the generator step (A/B switch)
ADAM_G_BESIDE = E._exp_env('PATCHGAN_ADAM_G_BESIDE') != '0'      # ... and G's Adam update behind that fork (A/B switch)
# (under data parallelism too -- with the runtime's default of 4 hardware queues it measured SLOWER there, 9.06 vs 8.89 ms with a one-rank
#  RCCL group: the compute stream shared a hardware queue with a stream waiting for the deferred pass; patchgan_amd/__init__.py asks
#  for 8 queues: 8.51)
DEFER_D_BWD_DP = E._exp_env('PATCHGAN_DEFER_D_BWD_DP') != '0'
DEFER_D_BWD = E._exp_env('PATCHGAN_DEFER_D_BWD') != '0'      # two-stream step: the discriminator's backward pass + Adam(D) under the NEXT step's generator forward (A/B switch)


_GC_FREEZES = 0


def _settle_gc(step):
    """Keep CPython's cyclic collector out of the step loop's way -- OPT-IN (Trainer.gc_freeze; bench.py and the patchgan_train
    entry point set it, a program that embeds Trainer is left alone: gc.freeze() is process-global and would move the host
    application's own objects to the permanent generation as well).  A full (generation-2) collection walks every live container
    object of the process -- ~0.5 M after `import torch` -- and takes 40-70 ms here: four to fifteen G+D steps of GPU time during
    which the host enqueues nothing (measured: one such pause inside a 20-step bench window moved bf16 cfg2 from 4.3 to 6.2 ms per
    step).  After the first and after the third training step of the process -- by then the engines' kernel plans, workspaces and
    weight caches exist -- collect once and move everything alive into the permanent generation (gc.freeze()): later collections
    only look at what the steps themselves create.  Trainer.train() undoes it (gc.unfreeze()) when it returns.
    PATCHGAN_GC_FREEZE=0 switches it off everywhere."""
    global _GC_FREEZES
    if _GC_FREEZES >= 2 or step not in (1, 3) or os.environ.get('PATCHGAN_GC_FREEZE', '1') == '0':
        return
    import gc
    gc.collect()
    gc.freeze()
    _GC_FREEZES += 1


def _unsettle_gc():
    """Hand the objects frozen by _settle_gc back to the collector (end of Trainer.train)."""
    global _GC_FREEZES
    if _GC_FREEZES:
        import gc
        gc.unfreeze()
        _GC_FREEZES = 0


class StepLosses(dict):
    """The six loss scalars of one step (reference trainer.py:108-115 returns them as a dict of floats).  A dict whose content
    arrives with the step's device-to-host copy: Trainer.batch returns as soon as the step is enqueued and the FIRST access waits
    for that copy, so a caller that reads the values later (the epoch loop reads step i after enqueuing step i + 1, bench.py after
    its timed region) does not drain the GPU between steps (the end-of-step wait left it idle for 0.12-0.15 ms: the host needs
    that long to get the next step's first kernels out)."""

    KEYS = ('gen', 'gen_loss', 'gdisc', 'discr', 'discf', 'disc')

    def __init__(self, host, event):
        super().__init__()
        self._host, self._event = host, event

    def _fill(self):
        if self._host is not None:
            self._event.synchronize()
            v = self._host.numpy()
            seg, gdisc = np.float32(v[0]), np.float32(v[1])
            loss_real, loss_fake = np.float32(v[2]) * np.float32(2), np.float32(v[3]) * np.float32(2)
            gen_loss = seg + gdisc
            disc_loss = (loss_fake + loss_real) / np.float32(2)
            vals = [float(gen_loss), float(gen_loss), float(gdisc), float(loss_real), float(loss_fake), float(disc_loss)]
            self._host = self._event = None
            dict.update(self, zip(self.KEYS, vals))
        return self

    def __getitem__(self, k):
        return dict.__getitem__(self._fill(), k)

    def __iter__(self):
        return dict.__iter__(self._fill())

    def __len__(self):
        return dict.__len__(self._fill())

    def __contains__(self, k):
        return dict.__contains__(self._fill(), k)

    def __repr__(self):
        return dict.__repr__(self._fill())

    def __eq__(self, other):
        if isinstance(other, StepLosses):
            other._fill()
        return dict.__eq__(self._fill(), other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def keys(self):
        return dict.keys(self._fill())

    def values(self):
        return dict.values(self._fill())

    def items(self):
        return dict.items(self._fill())

    def get(self, k, default=None):
        return dict.get(self._fill(), k, default)

    def copy(self):
        return dict(self._fill())

    def __reduce__(self):
        return (dict, (dict(self._fill()),))


def weights_init(net, init_type='normal', scaling=0.02):
    """The reference's ``weights_init`` defines an inner function and never applies it (trainer.py:327-343):
    a no-op, so torch's default initialisation stays.  Kept as a no-op for parity."""
    return None


class Trainer:
    seg_alpha = 200
    loss_type = 'tversky'
    tversky_beta = 0.75
    tversky_gamma = 0.75

    neptune_config = None
    label_values = None      # label list of the dataset, only for the uint8 (device-side) input path of batch()
    gc_freeze = False        # opt-in: gc.collect() + gc.freeze() after training steps 1 and 3 (_settle_gc; process-global)
    # How a step is launched (decided per KIND of step: shapes, loss settings, precision / tuning, train / eval):
    #   graph        False | True | 'auto'  -- replay the steady-state TRAINING step from a captured hipGraph (_replay): 220 launches become
    #                one hipGraphLaunch (host 1.9 -> 0.1 ms per step).  Only single-GPU steps without dropout can be captured.
    #   two_streams  None | False | True | 'auto'  -- launch by launch with each backward pass's weight-gradient chain (fp32 networks) and
    #                the discriminator step's forward pass on a second stream (engine.Exec).  Independent of capture: dropout, evaluation
    #                passes and data-parallel steps take it too.  None = follows `graph` ('auto' with graph = 'auto', else off).
    # 'auto': a tournament per kind of step.  After one untimed warm step every way of launching that the settings allow -- one stream,
    # two streams, the captured graph -- runs for 2 x TRIAL_STEPS steps (the first half settles: allocator growth for the new pattern) while an event is recorded at the start of each step; a candidate's
    # score is the SHORTEST start-to-start period it reached (the step time as the caller sees it: host-bound or device-bound, whatever
    # binds; the minimum is robust against a loader hiccup or a GC pause in one of the steps), and the fastest candidate is kept.  All
    # candidates are the same computation, bit for bit, so the trials are ordinary training steps.  Measured at cfg2 / cfg4 two streams
    # win (8.3 vs 8.9 ms), at cfg1 too (2.49 ms against 2.92 for the graph: its 219 short kernels leave gaps a second chain fills), a
    # tiny network ends on the graph.  AUTO_FORCE (tests, experiments) decrees the outcome instead of measuring it.
    graph = False
    two_streams = None
    GRAPH_WARM_STEPS = 3     # graph = True: eager steps of a given kind before it is captured (kernel plans, weight-cache plans, workspaces settle)
    TRIAL_STEPS = 4          # 'auto': per candidate, this many settling steps and then this many timed ones
    GRAPH_TRIAL_RATIO = 1.25 # 'auto': the captured graph enters the tournament only if the one-stream step takes < this x its own host enqueue time
    AUTO_FORCE = None        # 'eager1' | 'eager2' | 'graph': 'auto' takes this outcome (where the settings allow it) after the warm steps
    #                          ('graph2' = the two-stream fork / join schedule INSIDE the captured step: by decree only, it is not a tournament
    #                          candidate -- the replay runs its branches no faster than the one-stream capture and slower than the eager
    #                          two-stream step: cfg1 2.96 vs 2.89 vs 2.40 ms, cfg2 8.33 vs 7.02; EXPERIMENTS.md round 6)
    MAX_GRAPHS = 2           # captured kinds of step kept (each holds its activations: ~4 GB at cfg2)
    MAX_KINDS = 16           # kinds of step whose launch decision is remembered
    _hwq_warned = False      # (process-wide: the late GPU_MAX_HW_QUEUES warning is given once)

    def __init__(self, generator, discriminator, savefolder, device='cuda'):
        # (networks that an earlier Trainer drove may still have a discriminator update in flight: complete it before anything here
        #  touches their weights -- the modules' apply() does that through the access hook)
        generator.apply(weights_init)
        discriminator.apply(weights_init)
        self.generator = generator
        self.discriminator = discriminator
        # (a discriminator update may be in flight on the second stream / in a collective when batch() returns: anything that reads the
        #  module's weights through its own surface completes it first.  Hooks chain: an earlier trainer's stays in front of this one's)
        import weakref
        me = weakref.ref(self)
        earlier = getattr(discriminator, '_access_hook', None)

        def hook():
            if earlier is not None:
                earlier()
            t = me()
            if t is not None:
                t.flush()
        discriminator._access_hook = hook
        self.device = device
        if savefolder[-1] != '/':
            savefolder += '/'
        self.savefolder = savefolder
        if not os.path.exists(savefolder):
            os.makedirs(savefolder, exist_ok=True)
        self.start = 1
        self.gen_lr = self.dsc_lr = 1e-3
        self._adam = None          # (gm, gv, dm, dv) flat moment buffers
        self._t_g = self._t_d = 0  # Adam step counts
        self._step = 0
        self._pending_d = None
        self._deferred = None      # operands of a discriminator backward pass still running on the second stream (flush())
        self.bucket_bytes = 32 << 20   # all-reduce bucket size under data parallelism (parallel.GradReducer)
        self._graphs, self._adam_dev = {}, None
        self._kinds, self.step_times, self.launch_mode = {}, None, None      # per kind of step: warm-step count, the tournament, the decision; step_times = the last tournament's ms per step per candidate
        self._exec = None          # engine.Exec: this trainer's workspaces and second stream (created on the networks' device)
        self._oom_kinds, self.oom_fallbacks = set(), 0      # kinds of step pinned to one stream after an out-of-memory two-stream step

    # -------------------------------------------------------------------------------------- optimizers
    def setup_optimizers(self, gen_lr=1e-3, dsc_lr=1e-3):
        """Fresh Adam state (the reference re-creates both optimizers on every train() call, trainer.py:169-172)."""
        self.flush()              # an Adam(D) still running on the second stream reads and writes the OLD moments
        self.gen_lr, self.dsc_lr = gen_lr, dsc_lr
        g, d = self.generator.flat, self.discriminator.flat
        self._adam = (torch.zeros_like(g), torch.zeros_like(g), torch.zeros_like(d), torch.zeros_like(d))
        self._t_g = self._t_d = 0
        self._graphs, self._kinds = {}, {}        # captured steps update the OLD moment buffers

    # -------------------------------------------------------------------------------------- one G+D step
    def batch(self, x, y, train=False):
        t_host0 = time.perf_counter()
        G, D = self.generator, self.discriminator
        dev = G.flat.device
        if dev.type != 'cuda':
            raise RuntimeError("patchgan_amd.Trainer needs the networks on a HIP device (generator.to('cuda')); "
                               "there is no CPU path")
        # device-side input pipeline (opt-in, beyond the reference): decoded bytes x uint8 [N,H,W,Cin] + label map y uint8
        # [N,H,W]; `/255.` and the one-hot mask over self.label_values (io.py:42-56) run on the GPU after a 4x smaller H2D
        u8 = isinstance(x, torch.Tensor) and x.dtype == torch.uint8
        if u8:
            if self.label_values is None:
                raise RuntimeError("uint8 input needs Trainer.label_values (the dataset's label list)")
            x = x.to(dev, non_blocking=True)
            y = y.to(dev, non_blocking=True)
            N, H, W, Cin = x.shape
            Cout = len(self.label_values)
        else:
            if not isinstance(x, torch.Tensor):
                x = torch.as_tensor(np.asarray(x), dtype=torch.float)
                y = torch.as_tensor(np.asarray(y), dtype=torch.float)
            x = x.to(dev, dtype=torch.float32, non_blocking=True)
            y = y.to(dev, dtype=torch.float32, non_blocking=True)
            N, Cin, H, W = x.shape
            Cout = y.shape[1]
        ge, de = G.engine, D.engine
        if Cin != ge.input_nc or Cout != ge.output_nc or Cin + Cout != de.input_nc:
            raise RuntimeError(f"channel mismatch: x {Cin}, y {Cout} vs generator ({ge.input_nc}->{ge.output_nc}), "
                               f"discriminator input {de.input_nc}")
        if train and self._adam is None:
            self.setup_optimizers(self.gen_lr, self.dsc_lr)
        self._step += 1
        ex = self._exec
        if ex is None or ex.device != dev:
            self.flush()          # (a deferred pass is joined on the Exec it was enqueued on, before that one is let go)
            ex = self._exec = E.Exec(dev)
        with ex:
            dims = (N, H, W, Cin, Cout)
            key = self._kind_key(x, y, u8, dims, train)
            mode = self._launch_mode(key, train)
            losses = None
            try:
                if mode in ('graph', 'graph2'):
                    losses = self._replay(key, x, y, u8, dims, two=(mode == 'graph2'))   # None: this runtime cannot capture the step
                    if losses is None:
                        mode = 'eager1'
                if losses is None:
                    ex.enabled = mode == 'eager2'
                    t_g0, t_d0 = self._t_g, self._t_d
                    try:
                        losses = self._enqueue_step(x, y, u8, N, H, W, Cin, Cout, train)
                    except torch.cuda.OutOfMemoryError:
                        # the two-stream step holds ~2x the one-stream step's memory (second workspace, the early discriminator forward
                        # beside the generator's saved activations): where it does not fit, this KIND of step runs on one stream from
                        # now on.  Safe to run again only while nothing of this step was committed (no optimizer update enqueued).
                        if mode != 'eager2' or _dist().on or (self._t_g, self._t_d) != (t_g0, t_d0):
                            raise
                        self._two_stream_oom(ex, key)
                        mode = 'eager1'
                        losses = self._enqueue_step(x, y, u8, N, H, W, Cin, Cout, train)
                    finally:
                        ex.enabled = False
            except BaseException:
                kind = self._kinds.get(key)
                if kind is not None:
                    kind['trial'] = None          # a step that raised must not score in a running tournament: it starts over
                raise
            self.launch_mode = mode
        if train and self.gc_freeze:
            _settle_gc(self._step)
        self.host_ms = (time.perf_counter() - t_host0) * 1e3       # host time to enqueue the whole step (bench.py reports it)
        kind = self._kinds.get(key)
        if kind is not None and kind.get('trial') is not None:
            h = kind['trial']['host']
            h[mode] = min(h.get(mode, 1e30), self.host_ms)
        return self._publish(losses)

    marks = None             # diagnostics (tools/step_phases.py): a list -> every _mark(name) records an event on the current stream

    def _mark(self, name):
        if self.marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.marks.append((name, ev, time.perf_counter()))

    def _enqueue_step(self, x, y, u8, N, H, W, Cin, Cout, train):
        """Every launch of one step on the current stream, from the device-resident inputs to the four loss scalars (returned as a
        device tensor).  Nothing in here reads the device; the only step-dependent HOST values are the dropout seed (self._step) and
        Adam's step count / learning rate (_adam_step) -- with dropout off and self._adam_dev set the sequence is the same from step
        to step, which is what _batch_graph captures."""
        G, D = self.generator, self.discriminator
        ge, de = G.engine, D.engine
        dev = G.flat.device
        dist = _dist()
        if dist.on and not Trainer._hwq_warned:
            Trainer._hwq_warned = True
            import patchgan_amd
            if patchgan_amd.HWQ_LATE:
                import warnings
                warnings.warn('patchgan_amd: HIP was initialised before `import patchgan_amd`, so GPU_MAX_HW_QUEUES=8 could not take effect; '
                              'the data-parallel step keeps five streams busy and loses ~0.6 ms per step at cfg2 with the default 4 hardware '
                              'queues -- import patchgan_amd first or export GPU_MAX_HW_QUEUES=8')
        Bglobal = N * dist.world
        Cd = Cin + Cout
        ex = E.cur_exec(dev)
        self._mark('start')

        # discriminator input buffer: samples [0,N) real = x|y, [N,2N) fake = x|G(x)   (trainer.py:65,96,98)
        # (bf16 networks read it through pg_pad8_bf16: 8-float pixels there, so that pass and the fill below move whole 16-byte pieces)
        ld_din = 8 if (4 < Cd < 8 and (de.algo & L.ALGO_MASK) == L.ALGO_BF16) else Cd
        din = E.View(torch.empty(2 * N * H * W * ld_din, dtype=torch.float32, device=dev), 0, ld_din, 2 * N, H, W, Cd)
        real, fake = din.samples(0, N), din.samples(N, N)
        if u8:
            real.channels(0, Cin).from_u8(x)
            fake.channels(0, Cin).from_u8(x)
            real.channels(Cin, Cout).from_labels(y, self.label_values)
        elif Cd <= 8:
            E.din_fill(x, y, real, fake)       # both halves' x | y (and the fake half's zeroed mask channels) in one launch
        else:
            real.channels(0, Cin).from_nchw(x)
            fake.channels(0, Cin).from_nchw(x)
            real.channels(Cin, Cout).from_nchw(y)
        xin, yv, gen = fake.channels(0, Cin), real.channels(Cin, Cout), fake.channels(Cin, Cout)

        losses = torch.empty(4, dtype=torch.float32, device=dev)   # seg, gdisc, real*0.5, fake*0.5 (each written by its loss kernel)
        allred = dist.all_reduce_side if dist.on else None

        # ---- generator step
        seed = E._mix_seed(G._seed_base, self._step) if G.training else 0
        # G's weights are constant from here to its Adam step: its Winograd-transformed / packed bf16 weights for this step are made by
        # one batched launch (from the second step on; engine._WeightPrep), shared by forward and backward where one copy serves both
        gcache = ge.ucache_begin(G.flat, N, H, W) if train else None
        gc = ge.forward(G.flat, xin, gen, G.training, seed, sample0=dist.rank * N, keep_v=train, ucache=gcache)   # trainer.py:63
        self._mark('G fwd done')
        self.flush()          # D's deferred all-reduce + Adam from the previous step ran under this G forward
        self._mark('flushed')
        # seg loss, phase 1 (per-sample reductions); under data parallelism its two batch-global terms are summed across
        # ranks on the comm stream while the discriminator's forward pass over the fake batch runs
        seg_pending = E.loss_begin(gen, yv, 0.0, self.tversky_beta, allred)
        # D's weights do not change until the Adam step at the end of this call: its Winograd-transformed weights are computed
        # once per (layer, direction) and shared by the passes below through this per-step cache
        ucache = de.ucache_begin(D.flat, N, H, W, tag=bool(train))
        # two-stream step: the discriminator step's forward over din[2N] (trainer.py:96-99) needs only the generator's OUTPUT and D's
        # weights, both final here -- it is enqueued on the second stream now and runs under the rest of the generator step (same
        # kernels, same 2N plan: bit-identical); joined where the discriminator step reads its output
        dc2 = None
        # (evaluation passes and data-parallel steps too; under data parallelism D's deferred update of the previous step has been
        #  applied by flush() above, and on_side() orders the second stream behind it)
        if ex.enabled and E.PROFILER is None and EARLY_D_FWD:      # (bf16 networks too: 5.93 -> 5.84 ms at cfg4)
            n_prepared = len(ucache)
            with E.on_side():
                dc2 = de.forward(D.flat, din, ucache=ucache, keep_v=train)
                self._mark('side: dc2 fwd done')
            if len(ucache) != n_prepared:
                # (a transformed-weight entry the batched preparation did not cover was made by a kernel on the second stream -- the
                #  first steps after a change of tuning: the passes below would take it as ready, so they wait for it this once)
                E.side_join()
        dc = de.forward(D.flat, fake, ucache=ucache)                                          # trainer.py:66
        gseg = E.View.alloc(N, H, W, Cout, dev) if train else None
        E.loss_finish(seg_pending, _LOSS_MODES[self.loss_type], float(self.seg_alpha), gseg, losses, 0, Bglobal,
                      self.tversky_gamma)                                                     # trainer.py:71-82
        o = dc.out
        gd = E.View.alloc(o.N, o.H, o.W, 1, dev) if train else None
        E.loss_value_and_grad(o, None, 1.0, L.LOSS_BCE, 1.0, gd, losses, 1, Bglobal)          # trainer.py:84
        g_reducer, late_adam_g = None, False
        self._mark('D(fake) fwd + losses done')
        if train:
            gflat = G.ensure_grad_flat()
            ddin = de.backward(D.flat, None, dc, gd, need_wgrad=False, need_dx=True, ucache=ucache)   # dL/d(x|gen)
            self._mark('D dgrad done')
            if dist.on:
                # buckets of the flat G gradient are all-reduced on RCCL's stream as backward finishes them; the
                # collective keeps running under the discriminator step below (which does not read G's new weights:
                # gen_img.detach() is the pre-update output, trainer.py:98)
                # (two-stream step: a bucket's weight gradients may sit on the second stream -- the collective waits for it as well)
                g_reducer = GradReducer(dist, gflat, self.bucket_bytes, producers=E.side_producers)
            # two-stream step: G's weight gradients may still be running on the second stream when the pass returns; the
            # discriminator's forward below needs neither them nor G's new weights (gen_img.detach() is the pre-update output,
            # trainer.py:98), so the join and G's Adam update come after it
            two = bool(ex.enabled)
            late_adam_g = two and g_reducer is None
            ge.backward(G.flat, gflat, gc, gseg, ddin.channels(Cin, Cout),                    # trainer.py:88-89
                        on_ready=g_reducer.ready if g_reducer is not None else None, ucache=gcache, defer_join=two)
            ge.ucache_end(gcache)
            self._mark('G bwd (main chain) done')
            if late_adam_g:
                pass
            elif g_reducer is None:
                self._adam_step('g')                                                          # trainer.py:90
            else:
                g_reducer.finish(launch_only=True)     # the last (first layers') bucket leaves as soon as backward has produced it
        del dc

        # ---- discriminator step: real and (pre-update, detached) fake in one 2N batch        trainer.py:96-99
        if dc2 is None:
            dc2 = de.forward(D.flat, din, ucache=ucache, keep_v=train)
        if ex.pending or ex.keep:
            E.side_join()
        self._mark('joined')
        capturing = self._adam_dev is not None      # (a step being captured ends with every chain joined: nothing is deferred across its end)
        adam_g_behind_fork = bool(train and late_adam_g and ex.enabled and E.PROFILER is None and DEFER_D_BWD and ADAM_G_BESIDE and not capturing)
        if train and late_adam_g and not adam_g_behind_fork:
            self._adam_step('g')                                                              # trainer.py:90
        o2 = dc2.out
        god = E.View.alloc(o2.N, o2.H, o2.W, 1, dev) if train else None
        E.loss_value_and_grad(o2.samples(0, N), None, 1.0, L.LOSS_BCE, 0.5, god.samples(0, N) if train else None,
                              losses, 2, Bglobal)                                             # loss_real
        E.loss_value_and_grad(o2.samples(N, N), None, 0.0, L.LOSS_BCE, 0.5, god.samples(N, N) if train else None,
                              losses, 3, Bglobal)                                             # loss_fake
        wait_losses = None
        if dist.on:
            # the step's loss scalars for logging: seg is already global for tversky, the BCE / MAE terms are per-rank partial
            # means.  Summed on the comm stream under the discriminator's backward pass, ahead of its gradient all-reduce.
            if self.loss_type == 'tversky':
                losses[0] /= dist.world
            wait_losses = dist.all_reduce_side(losses)
        if train:
            dflat = D.ensure_grad_flat()
            if ex.enabled and E.PROFILER is None and DEFER_D_BWD and (g_reducer is None or DEFER_D_BWD_DP) and not capturing:
                # two-stream step: the discriminator's whole backward pass and its Adam update go to the second stream and run under the
                # NEXT step's generator forward, which reads neither D's weights nor its gradients (trainer.py:63; the next use of D is
                # trainer.py:66) and has no second chain of its own -- the same kernels with the same arguments, so the results are
                # bit-identical; flush() joins before anything reads D's weights (the next step's D passes, save / load, the end of train,
                # state_dict() / forward of the module).  The pass's operands stay referenced until then.
                with E.on_side():
                    de.backward(D.flat, dflat, dc2, god, need_wgrad=True, need_dx=False, ucache=ucache)   # trainer.py:106
                    if g_reducer is not None:
                        self._pending_d = dist.all_reduce_side(dflat)      # (behind the pass on the second stream; Adam(D) in flush())
                    else:
                        self._adam_step('d')                                                      # trainer.py:107
                    self._mark('side: D bwd done')
                self._deferred = (ex, dc2, god, ucache, dflat)
                if adam_g_behind_fork:
                    # G's Adam update (1.2 GB at the memory rate, 0.2 ms alone on the chip) beside the first kernels of D's backward pass
                    # instead of in front of them: it only has to precede the next step's generator forward
                    self._adam_step('g')                                                          # trainer.py:90
                if g_reducer is not None:
                    self._mark('forked')
                    g_reducer.finish()                     # G's buckets have been in flight since the generator backward
                    self._mark('G buckets waited')
                    self._adam_step('g')
                    self._mark('Adam(G) done')
            else:
                de.backward(D.flat, dflat, dc2, god, need_wgrad=True, need_dx=False, ucache=ucache)   # trainer.py:106
                if g_reducer is not None:
                    # D's gradient (11 MB at ndf=64) is all-reduced asynchronously and applied by flush() at the first use of
                    # D's weights -- after the NEXT step's generator forward, which does not read them (trainer.py:63-66); handed to
                    # the comm stream BEFORE the compute stream waits for G's buckets, so it is queued behind them without a gap
                    self._pending_d = dist.all_reduce_side(dflat)
                    g_reducer.finish()                     # G's buckets have been in flight since the generator backward
                    self._adam_step('g')
                else:
                    self._adam_step('d')                                                          # trainer.py:107
        de.ucache_end(ucache)
        if wait_losses is not None:
            wait_losses()
        self._mark('end')
        self._last_gen = gen
        return losses

    def _publish(self, losses):
        # the step's one device-to-host copy, asynchronous into a pinned slot: the returned dict waits for it on first access
        ring = getattr(self, '_loss_ring', None)
        if ring is None or ring[0][0].device != torch.device('cpu'):
            ring = self._loss_ring = [[torch.empty(4, dtype=torch.float32).pin_memory(), None] for _ in range(4)]
            self._loss_slot = 0
        slot = ring[self._loss_slot]
        self._loss_slot = (self._loss_slot + 1) % len(ring)
        if slot[1] is not None:
            slot[1]._fill()                # a result four steps old that nobody read yet: take its values out of the slot first
        slot[0].copy_(losses, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        slot[1] = out = StepLosses(slot[0], ev)
        return out

    def flush(self):
        """Complete a discriminator update that is still in flight: under data parallelism its gradient all-reduce (Adam(D) is applied
        here), in a two-stream step the backward pass + Adam(D) running on the second stream under the next generator forward.  Called
        before anything reads the discriminator's weights: the next step's D passes, save(), load(), the end of train(), and -- through
        the modules' access hook -- state_dict() / forward() / .to() of the discriminator itself."""
        if self._deferred is not None:
            ex = self._deferred[0]          # the execution state the pass was enqueued on (self._exec may have been replaced since)
            if ex is not None and (ex.pending or ex.keep):
                with ex:
                    E.side_join()
            self._deferred = None
        wait = getattr(self, '_pending_d', None)
        if wait is not None:
            self._pending_d = None
            wait()
            self._adam_step('d')

    def _adam_step(self, which):
        net, (m, v) = (self.generator, self._adam[0:2]) if which == 'g' else (self.discriminator, self._adam[2:4])
        if self._adam_dev is not None:
            # a step being captured: lr / bc1 and sqrt(bc2) come from device memory (written before every replay, _batch_graph)
            E.adam_step_dev(net.flat, net.grad_flat, m, v, self._adam_dev[0:2] if which == 'g' else self._adam_dev[2:4])
        elif which == 'g':
            self._t_g += 1
            E.adam_step(net.flat, net.grad_flat, m, v, self._t_g, self.gen_lr)
        else:
            self._t_d += 1
            E.adam_step(net.flat, net.grad_flat, m, v, self._t_d, self.dsc_lr)

    # -------------------------------------------------------------------------------------- how a step is launched
    def _graph_eligible(self):
        """Capture applies: single GPU, dropout off (its per-layer seeds are launch arguments derived from the step number on the
        host), and no launch profiler armed (its HIP events are per launch: the HIP runtime torch brings along refuses external
        event-record nodes inside a capture, tools/graph_event_probe.py)."""
        G = self.generator
        if not self.graph or os.environ.get('PATCHGAN_GRAPH', '1') == '0' or _dist().on or E.PROFILER is not None:
            return False
        return not (G.training and G.engine.use_dropout)

    def _two_streams_setting(self):
        ts = self.two_streams
        if ts is None:
            ts = 'auto' if self.graph == 'auto' else False
        if E.PROFILER is not None or E._exp_env('PATCHGAN_TWO_STREAMS') == '0':
            return False          # (the launch profiler's event pairs keep everything on one stream)
        return ts

    def _kind_key(self, x, y, u8, dims, train):
        G, D = self.generator, self.discriminator
        return (bool(train), u8, dims, self.loss_type, float(self.seg_alpha), float(self.tversky_beta), float(self.tversky_gamma),
                tuple(self.label_values) if self.label_values is not None else None, G.training, D.training,
                G.engine.algo, bool(G.engine.act_bf), D.engine.algo, bool(D.engine.act_bf), G.flat.data_ptr(), D.flat.data_ptr(),
                tuple(x.shape), tuple(y.shape), x.dtype, y.dtype, _dist().on)

    def _launch_mode(self, key, train):
        """'eager1' (launch by launch, one stream) | 'eager2' (launch by launch, two streams) | 'graph' (replay) for this step of kind
        `key`.  What the settings allow is re-read every step (a profiler armed later, a group initialised later); what was measured
        is remembered per kind."""
        want_graph = self.graph if (train and self._graph_eligible()) else False
        ts = self._two_streams_setting()
        if key in self._oom_kinds:
            ts = False                            # (did not fit in device memory: _two_stream_oom)
        if not want_graph and not ts:
            return 'eager1'
        if ts is True and want_graph is not True:
            return 'eager2'                       # forced: no warm-up needed (entries a pass makes on the second stream are joined, _enqueue_step)
        k = self._kinds.pop(key, None) or {'seen': 0, 'mode': None, 'trial': None}
        self._kinds[key] = k                      # most recently used last
        while len(self._kinds) > self.MAX_KINDS:
            self._kinds.pop(next(iter(self._kinds)))
        if k['mode'] is not None:
            if (k['mode'] in ('graph', 'graph2') and not want_graph) or (k['mode'] in ('eager2', 'graph2') and not ts):
                return 'eager1'
            return k['mode']
        k['seen'] += 1
        if want_graph is True:                    # capture, no questions: after the warm steps
            if k['seen'] <= self.GRAPH_WARM_STEPS:
                return 'eager1'
            k['mode'] = 'graph'
            return 'graph'
        cands = ['eager1'] + (['eager2'] if ts else []) + (['graph'] if want_graph else [])
        forced = os.environ.get('PATCHGAN_AUTO_FORCE') if 'PATCHGAN_EXPERIMENT' in os.environ else None
        forced = forced or self.AUTO_FORCE
        if forced is not None:                    # by decree (tests, A/B runs): after the warm steps, no trials
            if k['seen'] <= self.GRAPH_WARM_STEPS:
                return 'eager1'
            k['mode'] = forced if (forced in cands or (forced == 'graph2' and 'graph' in cands and 'eager2' in cands)) else 'eager1'
            return k['mode']
        if k['seen'] == 1:
            return 'eager1'                       # untimed: kernel plans, weight-cache plans, workspaces
        tr = k['trial']
        if tr is None or tr['cands'] != cands:    # (the settings changed under a running tournament, or a step raised: start over)
            tr = k['trial'] = {'cands': cands, 'i': 0, 'starts': [], 'ms': {}, 'host': {}, 'step': -1}
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()                               # the start of this step on the compute stream
        if tr['starts'] and tr['step'] != self._step - 1:
            # a step of ANOTHER kind ran since this kind's last one (validation batches, a ragged last batch, an evaluation pass): the
            # period would span it -- this candidate's trial starts over
            tr['starts'] = []
        tr['step'] = self._step
        tr['starts'].append(ev)
        if len(tr['starts']) <= 2 * self.TRIAL_STEPS + 1:
            return tr['cands'][tr['i']]
        # 2 x TRIAL_STEPS + 1 starts of this candidate and the start of the step that follows them: its periods are known once that last
        # event has been reached (one wait per candidate, during warm-up).  The first TRIAL_STEPS periods do not score: they hold the
        # switch and the caching allocator's growth for the new launch pattern (a second stream allocates from a pool of its own:
        # device allocations synchronise -- seen at 2-3 x the settled step time for the first steps)
        ev.synchronize()
        st = tr['starts']
        tr['ms'][tr['cands'][tr['i']]] = min(st[j].elapsed_time(st[j + 1]) for j in range(self.TRIAL_STEPS, len(st) - 1))
        tr['i'] += 1
        tr['starts'] = [ev]
        if tr['i'] < len(cands) and cands[tr['i']] == 'graph' and tr['ms']['eager1'] > self.GRAPH_TRIAL_RATIO * tr['host'].get('eager1', 0.0):
            # a replay removes host time only: where the one-stream step is well clear of being host-bound the capture (which would
            # hold a second copy of the step's activations, ~4 GB at cfg2) cannot win and is not tried
            tr['ms']['graph'] = None
            tr['i'] += 1
        if tr['i'] < len(cands):
            return cands[tr['i']]
        dist = _dist()
        if dist.on:
            # every rank keeps the same way of launching: the slowest rank's time per candidate decides (ranks reach this point at the
            # same step: trials and their restarts depend on the step sequence only)
            dev = self.generator.flat.device
            t = torch.tensor([tr['ms'][m] if tr['ms'][m] is not None else 1e30 for m in cands], dtype=torch.float64,
                             device=dev if dist.backend == 'nccl' else 'cpu')
            dist.dist.all_reduce(t, op=dist.dist.ReduceOp.MAX)
            for m, v in zip(cands, t.tolist()):
                if tr['ms'][m] is not None:
                    tr['ms'][m] = v
        self.step_times = dict(tr['ms'], host_enqueue=dict(tr['host']))
        k['mode'] = min((m for m in cands if tr['ms'][m] is not None), key=lambda m: tr['ms'][m])
        k['trial'] = None
        for two in (False, True):                 # the losing captures' buffers go back to the allocator
            if k['mode'] != ('graph2' if two else 'graph'):
                self._graphs.pop((key, two), None)
        return k['mode']

    def _two_stream_oom(self, ex, key):
        """A two-stream step ran out of device memory before anything was committed: join and drop the second stream's state, give the
        workspaces and the allocator's cache back, and pin this kind of step to one stream (redecide() forgets the pin)."""
        import warnings
        ex.enabled = False
        try:
            E.side_join()
        except Exception:
            pass
        torch.cuda.synchronize(ex.device)
        self._deferred = None
        ex.release()
        torch.cuda.empty_cache()
        self._oom_kinds.add(key)
        k = self._kinds.get(key)
        if k is not None:
            k['mode'], k['trial'] = 'eager1', None
        self.oom_fallbacks += 1
        warnings.warn('patchgan_amd: the two-stream step ran out of device memory (it holds about twice what the one-stream step '
                      'does: INTEGRATION.md, "Memory"); this kind of step runs on one stream from here on')

    def __del__(self):
        # a discriminator update still running on the second stream references this trainer's tensors: join before they are freed
        try:
            if self._deferred is not None:
                self.flush()
        except Exception:
            pass

    def redecide(self):
        """Forget every launch decision and captured step (the next steps of each kind warm up, are timed and decided again)."""
        self.flush()
        self._kinds, self._graphs = {}, {}
        self._oom_kinds = set()

    def release(self):
        """Give this trainer's captured steps, workspaces and second stream back (they also go with the object)."""
        self.flush()
        self._graphs = {}
        if self._exec is not None:
            self._exec.release()

    def _replay(self, key, x, y, u8, dims, two=False):
        """The training step replayed from a captured hipGraph: ~220 launches become one hipGraphLaunch (host enqueue 1.9 ms -> well
        under 0.1 ms per step at cfg2; the step itself is unchanged -- same kernels, same arguments, bit-identical results).
        Captured on first use.  Per replay the host copies the inputs into the graph's input buffers, writes Adam's two
        step-dependent scalars per network (lr / bc1, sqrt(bc2): pg_adam_step_dev reads them from device memory) and launches.
        Returns None (after a warning) if this runtime cannot capture the step: the caller continues launch by launch."""
        self.flush()
        key = (key, bool(two))            # (a kind's one-stream and two-stream captures are different graphs)
        st = self._graphs.get(key)
        if st is not None and st.ptrs != self._graph_ptrs():
            # the gradient / moment buffers the capture was made with were replaced (a network moved, .grad reset): capture again
            del self._graphs[key]
            st = None
        if st is None:
            try:
                st = self._capture(key, x, y, u8, dims, two)
            except Exception as e:          # a runtime that cannot capture this step: launch by launch from here on, loudly
                import warnings
                warnings.warn(f'patchgan_amd: hipGraph capture of the training step failed ({type(e).__name__}: {e}); continuing launch by launch')
                self.graph = False
                return None
        else:
            self._graphs[key] = self._graphs.pop(key)          # most recently used last
        st.x.copy_(x, non_blocking=True)
        st.y.copy_(y, non_blocking=True)
        self._t_g += 1
        self._t_d += 1
        host = st.host[st.slot]
        st.slot = (st.slot + 1) % len(st.host)
        h = host.numpy()
        h[0:2] = E.adam_scalars(self._t_g, self.gen_lr)
        h[2:4] = E.adam_scalars(self._t_d, self.dsc_lr)
        st.scal.copy_(host, non_blocking=True)
        st.graph.replay()
        self._last_gen = st.gen
        return st.losses

    def _graph_ptrs(self):
        G, D = self.generator, self.discriminator
        return tuple(t.data_ptr() if t is not None else 0 for t in (G.grad_flat, D.grad_flat) + tuple(self._adam))

    def _capture(self, key, x, y, u8, dims, two=False):
        class _StepGraph:
            pass
        st = _StepGraph()
        dev = x.device
        G, D = self.generator, self.discriminator
        st.x, st.y = torch.empty_like(x), torch.empty_like(y)
        st.scal = torch.zeros(4, dtype=torch.float32, device=dev)
        # (a ring: the copy of slot i to the device may still be queued when the host prepares the next steps; _publish's ring of
        #  four loss slots keeps the host at most four steps ahead of the device)
        st.host, st.slot = [torch.zeros(4, dtype=torch.float32).pin_memory() for _ in range(8)], 0
        while len(self._graphs) >= self.MAX_GRAPHS:
            self._graphs.pop(next(iter(self._graphs)))
        G.ensure_grad_flat()
        D.ensure_grad_flat()
        st.graph = torch.cuda.CUDAGraph()
        self._adam_dev = st.scal
        ex = E.cur_exec(dev)
        try:
            # two: the fork / join schedule of the two-stream step inside the capture -- the second stream joins the capture through the
            # events the engine already uses (stream waits become graph dependencies) and every chain it runs is joined again before the
            # step ends; the one piece that crosses the step boundary, the discriminator's deferred backward pass, stays inside the step
            ex.enabled = bool(two)
            with torch.cuda.graph(st.graph, capture_error_mode='thread_local'):
                st.losses = self._enqueue_step(st.x, st.y, u8, *dims, True)
                if two and (ex.pending or ex.keep):
                    E.side_join()
                st.gen = self._last_gen
        finally:
            ex.enabled = False
            self._adam_dev = None
        # The graph has baked in the addresses of buffers it does not own: the workspace(s), the transformed / packed weight entries of
        # both networks, the flat gradients and Adam's moments.  It keeps every one of them alive -- a workspace that grows for a larger
        # extent, a weight-cache pool cleared by set_precision / set_tuning or trimmed by its plan limit replace THEIR reference, not this
        # one, so a replay never writes into memory the allocator has handed to someone else -- and it is captured again if the
        # buffers whose CONTENT matters outside the graph (gradients, moments) were replaced (_graph_ptrs).
        st.hold = (E.cur_exec(dev).buffers() + list(G.engine.__dict__.get('_upool', {}).values())
                   + list(D.engine.__dict__.get('_upool', {}).values()) + [G.grad_flat, D.grad_flat] + list(self._adam))
        st.ptrs = self._graph_ptrs()
        self._graphs[key] = st
        return st

    def graph_captured(self):
        """True once some kind of training step runs from a captured graph (bench.py warms up until then)."""
        return bool(self._graphs)

    def graph_decided(self):
        """True once a kind of step has been captured or its launch mode decided ('auto')."""
        return bool(self._graphs) or any(k['mode'] is not None for k in self._kinds.values())

    def decided_modes(self):
        """The launch modes decided so far, one per kind of step, most recently used last ('eager1' | 'eager2' | 'graph')."""
        return [k['mode'] for k in self._kinds.values() if k['mode'] is not None]

    # -------------------------------------------------------------------------------------- epoch driver
    def train(self, train_data, val_data, epochs, dsc_learning_rate=1.e-3, gen_learning_rate=1.e-3, save_freq=10,
              lr_decay=None, decay_freq=5, reduce_on_plateau=False):
        """Reference trainer.py:117-279: Adam(lr, betas=(0.9, 0.999)) for G and D, optional exponential LR decay every
        `decay_freq` epochs (resumed as lr*decay^((start-1)/decay_freq)), per-epoch train + validation loops, checkpoint
        every `save_freq` epochs.  Returns (G_loss_ep, D_loss_ep): per-epoch means of the training losses."""
        if (lr_decay is not None) and not reduce_on_plateau:
            gen_lr = gen_learning_rate * (lr_decay) ** ((self.start - 1) / (decay_freq))
            dsc_lr = dsc_learning_rate * (lr_decay) ** ((self.start - 1) / (decay_freq))
        else:
            gen_lr, dsc_lr = gen_learning_rate, dsc_learning_rate

        if self.neptune_config is not None:
            self.neptune_config['model/parameters/gen_learning_rate'] = gen_lr
            self.neptune_config['model/parameters/dsc_learning_rate'] = dsc_lr
            self.neptune_config['model/parameters/start'] = self.start
            self.neptune_config['model/parameters/n_epochs'] = epochs

        self.setup_optimizers(gen_lr, dsc_lr)
        plateau = None
        if reduce_on_plateau:
            plateau = [_Plateau(gen_lr), _Plateau(dsc_lr)]
            if self.neptune_config is not None:
                self.neptune_config['model/parameters/scheduler'] = 'ReduceLROnPlateau'
        elif lr_decay is not None and self.neptune_config is not None:
            self.neptune_config['model/parameters/scheduler'] = 'ExponentialLR'
            self.neptune_config['model/parameters/decay_freq'] = decay_freq
            self.neptune_config['model/parameters/lr_decay'] = lr_decay

        history = {'gen': [], 'disc': []}
        try:
            return self._epochs(train_data, val_data, epochs, save_freq, lr_decay, decay_freq, plateau, history)
        finally:
            # also on an exception / KeyboardInterrupt inside the loop.  gc.unfreeze() is process-global: it also thaws what the host
            # application froze itself (which is why gc_freeze is opt-in)
            if self.gc_freeze:
                _unsettle_gc()

    def _epochs(self, train_data, val_data, epochs, save_freq, lr_decay, decay_freq, plateau, history):
        for epoch in range(self.start, epochs + 1):
            if _dist().rank == 0:
                print(f"Epoch {epoch} -- lr: {self.gen_lr:5.3e}, {self.dsc_lr:5.3e}")
                print("-------------------------------------------------------")
            train_mean = self._run_epoch(train_data, True, epoch, 'Training: ', dynamic_ncols=True)
            for key in history:
                history[key].append(train_mean[key])
            # the reference keeps ONE running-mean dict across both loops: a validation pass without batches reports the
            # training means (trainer.py:204-262)
            val_mean = dict(train_mean, **self._run_epoch(val_data, False, epoch, 'Validation: '))
            if self.neptune_config is not None:
                for phase, mean in (('train', train_mean), ('eval', val_mean)):
                    self.neptune_config[f'{phase}/gen_loss'].append(mean['gen'])
                    self.neptune_config[f'{phase}/disc_loss'].append(mean['disc'])

            if plateau is not None:
                self.gen_lr = plateau[0].step(val_mean['gen'])
                self.dsc_lr = plateau[1].step(val_mean['disc'])
            elif lr_decay is not None and epoch % decay_freq == 0:
                self.gen_lr *= lr_decay        # ExponentialLR.step() (trainer.py:266-270)
                self.dsc_lr *= lr_decay

            if epoch % save_freq == 0:
                self.save(epoch)
        self.flush()
        return history['gen'], history['disc']

    def _run_epoch(self, data, train, epoch, desc, **bar_kwargs):
        """One pass over `data` through batch(train=...): networks switched to train() / eval(), the data source
        reshuffled (a `shuffle()` method as the reference calls it, trainer.py:206,236, or a DistributedSampler's
        set_epoch under data parallelism), running means of the six loss scalars shown on the progress bar.
        Returns {key: mean over the pass}."""
        for net in (self.generator, self.discriminator):
            net.train(train)
        bar = tqdm.tqdm(data, desc=desc, disable=_dist().rank != 0, **bar_kwargs)
        if hasattr(data, 'shuffle'):
            data.shuffle()
        sampler = getattr(data, 'sampler', None)
        if hasattr(sampler, 'set_epoch'):
            sampler.set_epoch(epoch)           # otherwise every epoch repeats the first permutation / rank shards
        sums, count, mean = defaultdict(float), 0, {}

        def account(step_losses):
            nonlocal count, mean
            count += 1
            for key, value in step_losses.items():          # (waits for that step's device-to-host copy)
                sums[key] += value
            mean = {key: total / count for key, total in sums.items()}
            bar.set_postfix_str(" ".join(f"{key}: {value:.2e}" for key, value in mean.items()))

        pending = None          # the bar shows step i once step i + 1 is enqueued: the GPU never waits for the logging
        for input_img, target_mask in bar:
            cur = self.batch(input_img, target_mask, train=train)
            if pending is not None:
                account(pending)
            pending = cur
        if pending is not None:
            account(pending)
        return mean

    # -------------------------------------------------------------------------------------- checkpoints
    def save(self, epoch):
        """generator_ep_%03d.pth / discriminator_ep_%03d.pth holding torch-layout state_dicts (trainer.py:281-287)."""
        self.flush()
        if _dist().rank != 0:
            return
        gen_savefile = f'{self.savefolder}/generator_ep_{epoch:03d}.pth'
        disc_savefile = f'{self.savefolder}/discriminator_ep_{epoch:03d}.pth'
        print(f"Saving to {gen_savefile} and {disc_savefile}")
        torch.save(self.generator.state_dict_contiguous(), gen_savefile)
        torch.save(self.discriminator.state_dict_contiguous(), disc_savefile)

    def load_last_checkpoint(self):
        gen_ck = sorted(glob.glob(self.savefolder + "generator_ep*.pth"))
        dsc_ck = sorted(glob.glob(self.savefolder + "discriminator_ep*.pth"))
        gen_epochs = set(int(os.path.basename(c).replace('generator_ep_', '')[:-4]) for c in gen_ck)
        dsc_epochs = set(int(os.path.basename(c).replace('discriminator_ep_', '')[:-4]) for c in dsc_ck)
        try:
            assert len(gen_epochs) > 0, "No checkpoints found!"
            start = max(gen_epochs.union(dsc_epochs))
            self.load(f"{self.savefolder}/generator_ep_{start:03d}.pth", f"{self.savefolder}/discriminator_ep_{start:03d}.pth")
            self.start = start + 1
        except Exception as e:
            print(e)
            print("Checkpoints not loaded")

    def load(self, generator_save, discriminator_save):
        self.flush()
        print(generator_save, discriminator_save)
        dev = self.generator.flat.device
        self.generator.load_state_dict(torch.load(generator_save, map_location=dev))
        self.discriminator.load_state_dict(torch.load(discriminator_save, map_location=dev))
        print(f"Loaded checkpoints from {os.path.basename(generator_save)} and {os.path.basename(discriminator_save)}")


class _Plateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau with its defaults, as the reference constructs it (trainer.py:175-178: mode
    'min', factor 0.1, patience 10, relative threshold 1e-4, cooldown 0, min_lr 0, eps 1e-8), on a scalar learning rate:
    step(metric) after every epoch (trainer.py:271-273) returns the learning rate for the next one.
    tests/test_trainer_cpu.py holds it against torch's class."""

    def __init__(self, lr):
        self.lr, self.best, self.bad = lr, float('inf'), 0

    def step(self, metric):
        metric = float(metric)
        if metric < self.best * (1 - 1e-4):
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
        if self.bad > 10:
            new_lr = max(self.lr * 0.1, 0.0)
            if self.lr - new_lr > 1e-8:        # torch ignores updates smaller than eps
                self.lr = new_lr
            self.bad = 0
        return self.lr
