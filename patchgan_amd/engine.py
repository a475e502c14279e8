"""Device engines for the patchGAN hot path: UNet generator and PatchGAN discriminator forward /
backward scheduled as explicit HIP kernel launches through the C ABI (include/patchgan_hip.h).

Data layout in HBM
  * activations: fp32 NHWC.  A `View` is (tensor, element offset, pixel stride ld, N, H, W, C): a channel
    slice of a wider buffer.  Skip connections (reference unet.py:127 ``torch.cat([x, xencs[i]], dim=1)``)
    are never materialised by a copy: decoder level i owns one buffer ``cat_i`` of C_dec + C_skip channels;
    decoder block i-1 writes channels [0, C_dec) and encoder block 6-i writes channels [C_dec, ...) directly.
    The discriminator input cat(x, mask) (trainer.py:65,96,98) is handled the same way.
  * weights: ONE flat fp32 buffer per network holding, per layer, the packed block P[16][a][b] (+ bias).  The
    torch-layout tensor W[a][b][4][4] the reference's state_dict exposes is a *strided view* of that block
    (strides (b, 1, 4ab, ab)), so checkpoints interchange with the reference without any repack kernel, and
    Adam / RCCL all-reduce run over the flat buffer in one launch.

Nothing here touches autograd; `patchgan_amd.unet` / `.disc` wrap the engines in torch.autograd.Function and
`patchgan_amd.trainer` drives them directly.
"""
import ctypes
import os
import math
import threading

import numpy as np

import torch

from . import _lib as L

_MASK64 = (1 << 64) - 1
# PATCHGAN_ALGO=direct forces the one-thread-per-output kernels everywhere (debugging aid); =bf16 selects the bf16
# MFMA variants (fp32 tensors, bf16 multiply, fp32 accumulate); default = the fp32 MFMA path
DEFAULT_ALGO = {'direct': L.ALGO_DIRECT, 'mfma': L.ALGO_MFMA, 'bf16': L.ALGO_BF16}.get(__import__('os').environ.get('PATCHGAN_ALGO', ''), L.ALGO_AUTO)


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    # the raw handle of torch's current stream on the current device, without building a torch.cuda.Stream object per launch
    # (torch.cuda.current_stream() was a quarter of the step's host time: 170 calls per step, tools/host_profile.py)
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


class View:
    """NHWC channel-slice of a device buffer; `bf` = the buffer holds bf16 elements (bf16 activation storage), else fp32.
    `off` and `ld` count elements of the buffer's own type."""
    __slots__ = ('t', 'off', 'ld', 'N', 'H', 'W', 'C', 'bf')

    def __init__(self, t, off, ld, N, H, W, C, bf=False):
        self.t, self.off, self.ld, self.N, self.H, self.W, self.C, self.bf = t, off, ld, N, H, W, C, bf

    @staticmethod
    def alloc(N, H, W, C, device, zero=False, bf=False):
        t = (torch.zeros if zero else torch.empty)(N * H * W * C, dtype=torch.bfloat16 if bf else torch.float32, device=device)
        return View(t, 0, C, N, H, W, C, bf)

    def ptr(self):
        return self.t.data_ptr() + self.off * (2 if self.bf else 4)

    def padded8(self):
        """This fp32 view (C <= 8) as a bf16 tensor in 8-channel pixels (ld 8, pad channels zero): the layout in which the bf16
        kernels take the image-facing tensors (pg_pad8_bf16)."""
        assert not self.bf and self.C <= 8
        t = torch.empty(self.npix * 8, dtype=torch.bfloat16, device=self.t.device)
        L.check(L.load().pg_pad8_bf16(self.ptr(), self.ld, t.data_ptr(), self.npix, self.C, _stream()), 'pg_pad8_bf16')
        return View(t, 0, 8, self.N, self.H, self.W, self.C, True)

    def channels(self, c0, c):
        assert 0 <= c0 and c0 + c <= self.C
        return View(self.t, self.off + c0, self.ld, self.N, self.H, self.W, c, self.bf)

    def samples(self, n0, n):
        assert 0 <= n0 and n0 + n <= self.N
        return View(self.t, self.off + n0 * self.H * self.W * self.ld, self.ld, n, self.H, self.W, self.C, self.bf)

    def converted(self, bf):
        """A dense copy of this view in the other storage type (pg_act_fwd_t with PG_ACT_NONE), or self if it already has it."""
        if self.bf == bf:
            return self
        out = View.alloc(self.N, self.H, self.W, self.C, self.t.device, bf=bf)
        L.check(L.load().pg_act_fwd_t(self.ptr(), self.ld, out.ptr(), out.ld, self.npix, self.C, L.ACT_NONE, 0.0, 0, _stream(),
                                      (1 if self.bf else 0) | (2 if bf else 0)), 'pg_act_fwd_t')
        return out

    @property
    def HW(self):
        return self.H * self.W

    @property
    def extent_bytes(self):
        """Bytes from this view's first element to the end of its last pixel row (what a kernel's 32-bit offsets must span)."""
        return self.npix * self.ld * (2 if self.bf else 4)

    @property
    def npix(self):
        return self.N * self.H * self.W

    def to_nchw(self):
        """Materialise as a contiguous NCHW fp32 torch tensor (API edge)."""
        if self.bf:
            return self.converted(False).to_nchw()
        out = torch.empty(self.N, self.C, self.H, self.W, dtype=torch.float32, device=self.t.device)
        L.check(L.load().pg_nhwc_to_nchw(self.ptr(), self.ld, out.data_ptr(), self.N, self.C, self.H, self.W,
                                         _stream()), 'pg_nhwc_to_nchw')
        return out

    def from_nchw(self, src):
        """Fill from a contiguous NCHW torch tensor of shape [N, C, H, W]."""
        assert tuple(src.shape) == (self.N, self.C, self.H, self.W), (tuple(src.shape), (self.N, self.C, self.H, self.W))
        assert not self.bf, 'API-edge layout kernels write fp32 views'
        src = src.contiguous()
        L.check(L.load().pg_nchw_to_nhwc(src.data_ptr(), self.ptr(), self.ld, self.N, self.C, self.H, self.W,
                                         _stream()), 'pg_nchw_to_nhwc')
        return self

    def from_u8(self, src, div=255.0):
        """Fill from decoded image bytes: uint8 device tensor [N, H, W, C] -> value / div (reference io.py:42)."""
        assert src.dtype == torch.uint8 and tuple(src.shape) == (self.N, self.H, self.W, self.C), tuple(src.shape)
        src = src.contiguous()
        L.check(L.load().pg_u8_to_f32(src.data_ptr(), self.ptr(), self.ld, self.npix, self.C, div, _stream()), 'pg_u8_to_f32')
        return self

    def from_labels(self, src, labels, add=1):
        """Fill with the one-hot mask of a uint8 label map [N, H, W]: channel i = ((uint8)(src + add) == labels[i])
        (reference io.py:43,53-56)."""
        assert src.dtype == torch.uint8 and tuple(src.shape) == (self.N, self.H, self.W), tuple(src.shape)
        assert len(labels) == self.C, (len(labels), self.C)
        src = src.contiguous()
        arr = (ctypes.c_int * len(labels))(*[int(v) for v in labels])
        L.check(L.load().pg_labels_to_onehot(src.data_ptr(), self.ptr(), self.ld, self.npix, arr, len(labels), add, _stream()),
                'pg_labels_to_onehot')
        return self


def din_fill(x, y, real, fake):
    """real = x | y, fake = x | 0 (NHWC fp32 views of one pixel stride, x / y contiguous NCHW): the discriminator's concatenated inputs
    (trainer.py:65,96,98) in one launch; the generator's head writes its output into fake's mask channels later in the step."""
    N, Cx, H, W = x.shape
    Cy = y.shape[1]
    assert tuple(y.shape) == (N, Cy, H, W) and (real.N, real.H, real.W, real.C) == (N, H, W, Cx + Cy) == (fake.N, fake.H, fake.W, fake.C)
    assert real.ld == fake.ld and not real.bf and not fake.bf
    x, y = x.contiguous(), y.contiguous()
    L.check(L.load().pg_din_fill(x.data_ptr(), y.data_ptr(), real.ptr(), fake.ptr(), real.ld, N, Cx, Cy, H, W, _stream()), 'pg_din_fill')


# ------------------------------------------------------------------------------------------------
# execution state: workspaces (split-K slabs) and the second stream, owned by whoever drives the engines
# ------------------------------------------------------------------------------------------------
# Weight gradients on a second stream.  Inside a network's backward pass a layer's weight gradient and its data gradient both start
# from dy and are independent of each other; enqueued on one stream their kernels run strictly one after the other, and every kernel
# whose grid is one round of workgroups loses its ramp, its lock-step prologue / epilogue phases and its tail (EXPERIMENTS.md, "the fp32
# polyphase GEMM's tile life": ~15-20 % of such a launch).  With the weight-gradient chain on a second stream (its own workspace) the
# two chains' kernels fill each other's gaps: cfg2 fp32 8.85 -> 8.68 ms per step with only the encoder's and the discriminator's
# weight gradients moved.  Same kernels, same order of every floating-point sum: bit-identical results.  The operands are referenced
# until the streams join at the end of the backward pass (the caching allocator must not hand them out meanwhile).  Off under the
# launch profiler, for bf16 networks' weight gradients and inside a hipGraph capture; it costs ~0.6 ms of host time per step (26 stream
# hand-overs), so it pays where the device step is several times the host's enqueue time (the Trainer measures both and picks: cfg2
# fp32 8.95 -> 8.65 ms, cfg4 fp32 16.7 -> 16.5; cfg1's 2.9-ms step stays on the captured graph).


class Exec:
    """Execution state of ONE driver of the engines (a Trainer; or a device's default for stand-alone module calls): the split-K
    workspace per stream and the book-keeping of a two-stream step.  Nothing in here is shared between owners -- two Trainers (also on
    different GPUs, also on different threads) never see each other's flags, held operands or buffers, and everything is released with
    the owner (or by release()).  The second STREAM is the device's (one per device, _SECOND_STREAMS): owners on one device use it in
    order, each behind its own hand-overs.

      enabled   set by the owner per step: this step may use the second stream
      allow     inside a backward pass: weight gradients go to the second stream
      inside    what is enqueued right now goes to the second stream (its own workspace)
      pending   something was enqueued under on_side() and has not been joined
      keep      operands of second-stream launches, referenced until the join
      ws_gen    counts workspace (re)allocations: a captured graph holds the buffers it was captured with (Trainer._capture)"""

    def __init__(self, device=None):
        self.device = torch.device(device) if device is not None else None
        self.enabled = self.allow = self.inside = self.pending = False
        self.stream = None
        self.keep = []
        self.keep_ptrs = set()
        self.keep_bytes = 0
        self.ws = {}
        self.ws_gen = 0
        self._prev = []

    def side_stream(self):
        """The second stream of a two-stream step, created on first use ON THE OWNER'S DEVICE."""
        if self.stream is None:
            dev = self.device if self.device is not None else torch.device('cuda', torch.cuda.current_device())
            st = _SECOND_STREAMS.get(dev.index)
            if st is None:
                st = _SECOND_STREAMS[dev.index] = torch.cuda.Stream(device=dev)
            self.stream = st
        return self.stream

    def workspace(self, nbytes, device):
        if self.device is not None and device != self.device:
            raise RuntimeError(f'patchgan_amd: engine call on {device} under an execution state bound to {self.device}')
        ws = self.ws.get(self.inside)
        if ws is None or ws.numel() < nbytes:
            ws = self.ws[self.inside] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
            self.ws_gen += 1
        return ws

    def buffers(self):
        """The device buffers launches under this state write into besides their arguments (what a captured graph must keep alive)."""
        return list(self.ws.values())

    def release(self):
        """Give the workspaces back to the allocator (the second stream is joined first)."""
        if self.stream is not None and (self.keep or self.pending):
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
        self.keep.clear()
        self.keep_ptrs.clear()
        self.keep_bytes = 0
        self.pending = self.allow = self.inside = False
        self.ws.clear()
        self.ws_gen += 1

    def __enter__(self):
        self._prev.append(getattr(_TLS, 'cur', None))          # (a stack: flush() inside batch() enters the same state again)
        _TLS.cur = self
        return self

    def __exit__(self, *exc):
        _TLS.cur = self._prev.pop()
        return False


_TLS = threading.local()
# ONE second stream per device for every owner on it.  A stream is a device resource: torch hands out streams from a fixed pool and the HIP
# runtime maps each onto one of its hardware queues -- a process that went through several trainers (bench.py's legs) otherwise ended up
# with a second stream sharing the compute stream's hardware queue: the fourth trainer's two-stream step took 6.9 ms instead of 2.5.
# In-order sharing is harmless: every hand-over to / from it is a stream wait, and owners' flags, held operands and workspaces stay
# their own.
_SECOND_STREAMS = {}
_DEFAULT_EXEC = {}       # device index -> Exec of stand-alone module calls (UNet(...)(x), predict_image): one workspace, never a second stream


def cur_exec(device=None):
    """The execution state of the running driver (`with exec:`), or the device's default one for stand-alone module calls."""
    ex = getattr(_TLS, 'cur', None)
    if ex is not None:
        return ex
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device())
    ex = _DEFAULT_EXEC.get(device.index)
    if ex is None:
        ex = _DEFAULT_EXEC[device.index] = Exec(device)
    return ex


def release_workspaces():
    """Free the default execution states' workspaces (stand-alone module calls; a Trainer's go with the Trainer or by
    Trainer.release())."""
    for ex in _DEFAULT_EXEC.values():
        ex.release()
    _DEFAULT_EXEC.clear()


# the owner joins every KEEP_LIMIT_BYTES of operands held for the second stream (with defer_join nearly all backward
# temporaries of a pass were live at once; joining costs one stream wait, ~10 us)
KEEP_LIMIT_BYTES = 3 << 30


class on_side:
    """`with on_side():` -- what is enqueued inside goes to the second stream (after everything enqueued so far), with that stream's
    own workspace; the caller joins with side_join() before it reads the results on its own stream.  Tensors allocated inside belong
    to the second stream's allocator pool: safe as long as every later use on the second stream is again behind such a hand-over
    (it is: each one starts with wait_stream)."""
    def __enter__(self):
        ex = self._ex = cur_exec()
        side = ex.side_stream()
        side.wait_stream(torch.cuda.current_stream())
        self._ctx = torch.cuda.stream(side)
        self._ctx.__enter__()
        self._was = ex.inside
        ex.inside = ex.pending = True
        return side

    def __exit__(self, *exc):
        self._ex.inside = self._was
        return self._ctx.__exit__(*exc)


def _side_begin(allow):
    # (not inside a hipGraph capture: a captured two-branch step replays SLOWER than the one-stream one, 9.03 vs 8.96 ms at cfg2)
    ex = cur_exec()
    # (nor inside on_side(): a pass the caller has put on the second stream as a whole runs its chains one after the other there)
    ex.allow = (bool(allow) and ex.enabled and not ex.inside and WGRAD_SIDE and PROFILER is None
                and not torch.cuda.is_current_stream_capturing())


def side_join():
    """The caller's join: of a backward pass called with defer_join=True and of anything enqueued under on_side() (no-op if nothing
    is pending)."""
    _side_join(True)


def _side_join(explicit=False):
    """The weight gradients enqueued on the second stream are complete for the current stream; their operands may be freed.  (The
    join at the end of a backward pass does not wait for work the CALLER put on the second stream under on_side(): only the caller's
    own side_join() does.)"""
    ex = cur_exec()
    ex.allow = False
    if ex.keep or (explicit and ex.pending):
        torch.cuda.current_stream().wait_stream(ex.stream)
        ex.keep.clear()
        ex.keep_ptrs.clear()
        ex.keep_bytes = 0
        ex.pending = False


def side_producers():
    """Streams besides the current one that may still be writing weight gradients (data parallelism: the bucket all-reduce waits
    for them too, parallel.Dist.all_reduce_side)."""
    ex = cur_exec()
    return [ex.stream] if (ex.stream is not None and (ex.keep or ex.pending)) else []


def _workspace(nbytes, device):
    return cur_exec(device).workspace(nbytes, device)


class LaunchProfiler:
    """Optional per-launch timing of the conv kernels with HIP events on the launch stream (bench.py's roofline
    leg).  Enabled by assigning an instance to ``engine.PROFILER``; costs two event records per launch."""
    def __init__(self, only=None):
        self.records = []   # (symbol, split, flops, start_event, end_event)
        self.only = only    # symbol: time only this kernel's launches
        self.limit = None   # stop arming events after this many records (bench.py: a bounded sample of the timed region)
        self._pool = []

    def reserve(self, n):
        """Create n events ahead of the timed region: creating (and first-recording, which is what makes torch allocate the
        hipEvent_t) an event per launch inside it costs host time and a stream operation each."""
        for _ in range(n):
            self._pool.append(self._new_event())

    def _new_event(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()          # torch creates the hipEvent_t lazily, at the first record: force it (the C side re-records the event)
        return e

    def launch(self, op, opcode, fn, io=0):
        if op.algo & L.ALGO_MASK == L.ALGO_DIRECT:
            return fn()
        sym, split = op.describe(opcode, io)
        if (self.only is not None and sym != self.only) or (self.limit is not None and len(self.records) >= self.limit):
            return fn()
        e0, e1 = self._event(), self._event()
        # the C ABI records the pair tightly around the main GEMM kernel of this call (not its split-K reduce)
        L.check(L.load().pg_conv_time_next(e0.cuda_event, e1.cuda_event), 'pg_conv_time_next')
        fn()
        self.records.append((sym, split, op.flops, op.kernel_flops(opcode, io), op.useful_flops(opcode, io), e0, e1))

    def launch2(self, op, opcodes, fn, io=0):
        """A fused call with two main GEMM kernels (pg_conv4x4_bwd_big): one event pair per kernel."""
        if op.algo & L.ALGO_MASK == L.ALGO_DIRECT:
            return fn()
        syms = [op.describe(oc, io)[0] for oc in opcodes]
        if (self.only is not None and self.only not in syms) or (self.limit is not None and len(self.records) >= self.limit):
            return fn()
        ev = [self._event() for _ in range(4)]
        L.check(L.load().pg_conv_time_next2(*[e.cuda_event for e in ev]), 'pg_conv_time_next2')
        fn()
        for i, oc in enumerate(opcodes):
            if self.only is None or syms[i] == self.only:
                self.records.append((syms[i], op.describe(oc, io)[1], op.flops, op.kernel_flops(oc, io), op.useful_flops(oc, io), ev[2 * i], ev[2 * i + 1]))

    def _event(self):
        if self._pool:
            return self._pool.pop()
        return self._new_event()

    def summary(self):
        """{symbol: dict(launches, ms, flops, kflops, uflops)} over every launch of that kernel symbol (all split-K factors, like a
        rocprofv3 --stats row): flops = algorithmic (direct-convolution) FLOPs of the layers, kflops = FLOPs the kernel
        executed (fewer for Winograd kernels; ragged edge tiles count as whole tiles), uflops = the kernel's algorithm on the exact
        extents (no tile padding) -- call after a device synchronize."""
        out = {}
        for sym, split, flops, kflops, uflops, e0, e1 in self.records:
            ms = e0.elapsed_time(e1)
            d = out.setdefault(sym, dict(launches=0, ms=0.0, flops=0.0, kflops=0.0, uflops=0.0))
            d['launches'] += 1
            d['ms'] += ms
            d['flops'] += flops
            d['kflops'] += kflops
            d['uflops'] += uflops
        return out


PROFILER = None
_MAX_TENSOR_BYTES = None      # pg_conv_max_tensor_bytes(), read once


class ConvOp:
    """One 4x4 / padding-1 convolution layer bound to its geometry: the three kernels that touch its weights."""

    def __init__(self, N, Hb, Wb, Ca, Cb, stride, algo=L.ALGO_AUTO):
        Hs = (Hb - 2) // stride + 1
        Ws = (Wb - 2) // stride + 1
        if Hb < 2 or Wb < 2 or Hs < 1 or Ws < 1:
            raise RuntimeError(f"Kernel size can't be greater than actual input size ({Hb}x{Wb}, 4x4 kernel, pad 1)")
        self.g = L.ConvGeom(N, Hb, Wb, Hs, Ws, Ca, Cb, stride)
        self.N, self.Hb, self.Wb, self.Hs, self.Ws, self.Ca, self.Cb, self.stride = N, Hb, Wb, Hs, Ws, Ca, Cb, stride
        self.algo = algo
        lib = L.load()
        self.ws_bytes = max(int(lib.pg_conv_workspace_bytes(ctypes.byref(self.g), op)) for op in (0, 1, 2, 3))
        self._desc = {}

    @property
    def ws_arg(self):
        """The workspace size every query (pg_conv_kernel, pg_conv_u_bytes, pg_conv_stats_chunks, ...) AND every launch of this
        op passes to the C ABI: the SAME number on both sides, so a hand-over sized by a query can never meet a launch that
        planned with a different workspace (the device buffer itself is shared and may be larger)."""
        return max(self.ws_bytes, 1 << 20)

    def _ws(self, device):
        ws = _workspace(self.ws_bytes, device)          # (at least ws_arg bytes: _workspace never allocates less than 1 MiB)
        return ws.data_ptr(), self.ws_arg

    @property
    def flops(self):
        """Algorithmic FLOPs of any of the three kernels on this geometry: 2 * N*Hs*Ws * 16 * Ca * Cb."""
        return 2.0 * self.N * self.Hs * self.Ws * 16 * self.Ca * self.Cb

    def kernel_flops(self, opcode, io=0):
        """FLOPs the main GEMM kernel of this call really executes on the MFMA pipe: the algorithmic count for the implicit
        GEMM kernels, 2.25-4x fewer for the Winograd kernels (their tile counts include the ragged-edge padding)."""
        return self._describe(opcode, io)[2]

    def useful_flops(self, opcode, io=0):
        """The main GEMM kernel's FLOPs without the padding of ragged edge tiles to whole tiles (pg_conv_kernel_flops): equal to
        kernel_flops for the implicit-GEMM kernels, 6-27 % below it for the Winograd kernels at cfg2."""
        ex, us = ctypes.c_double(0), ctypes.c_double(0)
        L.check(L.load().pg_conv_kernel_flops(ctypes.byref(self.g), opcode + 16 * (self.algo | io), self.ws_arg, ctypes.byref(ex),
                                              ctypes.byref(us)), 'pg_conv_kernel_flops')
        return us.value

    def describe(self, opcode, io=0):
        """(kernel symbol, split-K factor) of the main GEMM kernel the C ABI launches for this op (pg_conv_kernel: the
        dispatch code itself reports it).  io: the PG_IO_* bits of the call (bf16 activation storage)."""
        return self._describe(opcode, io)[:2]

    def _describe(self, opcode, io=0):
        key = (opcode, io)
        if key not in self._desc:
            name = ctypes.create_string_buffer(128)
            s, fl = ctypes.c_int(0), ctypes.c_double(0)
            L.check(L.load().pg_conv_kernel(ctypes.byref(self.g), opcode + 16 * (self.algo | io), self.ws_arg, name, 128,
                                            ctypes.byref(s), ctypes.byref(fl)), 'pg_conv_kernel')
            self._desc[key] = (name.value.decode(), s.value, fl.value)
        return self._desc[key]

    @staticmethod
    def _aligned(*views):
        return all(v.ld % 4 == 0 and v.ptr() % 16 == 0 and not v.bf for v in views)     # (the Winograd hand-overs are fp32 paths)

    @staticmethod
    def fits(*views):
        """Every view is addressable by the buffer-load kernels (pg_conv_max_tensor_bytes: 32-bit byte offsets).  The C side's size
        queries see the geometry only; the launches also see the views and leave the fast / bf16 paths for a tensor beyond this
        limit -- so hand-overs (part, v_keep, u_cache, mul_t) are planned only for views that fit."""
        global _MAX_TENSOR_BYTES
        if _MAX_TENSOR_BYTES is None:
            _MAX_TENSOR_BYTES = int(L.load().pg_conv_max_tensor_bytes())
        return all(v.extent_bytes < _MAX_TENSOR_BYTES for v in views)

    @staticmethod
    def _io(big, small):
        return (L.IO_BIG_BF16 if big.bf else 0) | (L.IO_SMALL_BF16 if small.bf else 0)

    def _query(self, key, fn):
        if key not in self._desc:
            self._desc[key] = int(fn())
        return self._desc[key]

    def stats_chunks(self, opcode, view_in, view_out):
        """Partial-sum chunks per sample the kernel of big2small (0) / small2big (1) can emit next to its output (K5: InstanceNorm
        statistics from the conv epilogue); 0 = not on this path (or the views are not 16-byte aligned)."""
        big, small = (view_in, view_out) if opcode == 0 else (view_out, view_in)
        io = self._io(big, small)
        if not self.fits(view_in, view_out):
            return 0
        if io:      # bf16 tensors: the LDS-DMA kernels' STATS epilogue (both tensors bf16, 16-byte-aligned views)
            if io != L.IO_MASK or not all(v.ptr() % 16 == 0 and v.ld % 8 == 0 for v in (view_in, view_out)):
                return 0
        elif not self._aligned(view_out) or (not self._aligned(view_in) and not (opcode == 0 and self.Cb <= 3)):
            return 0          # (the image-facing kernel k_b2s_tapkp gathers <= 3-channel pixels by scalars: any input view)
        return self._query(('chunks', opcode, io), lambda: L.load().pg_conv_stats_chunks(ctypes.byref(self.g), opcode, self.algo | io,
                                                                                        self.ws_arg))

    def u_bytes(self, opcode, io=0):
        """Bytes of the transformed weights the kernel of big2small (0) / small2big (1) works from (0: no weight transform on
        this path): what a caller-owned cache for pg_conv_extras.u_cache must hold.  io: the PG_IO_* bits of the call (the bf16
        kernels on bf16 tensors work from a packed bf16 copy of the weights)."""
        return self._query(('u', opcode, io), lambda: L.load().pg_conv_u_bytes(ctypes.byref(self.g), opcode, self.algo | io,
                                                                               self.ws_arg))

    def v_bytes(self):
        """Bytes of the polyphase-transformed `big` tensor that big2small can keep (v_keep) for the weight gradient of the
        same layer (v_pre); 0 when the two calls do not both take the polyphase Winograd path."""
        return self._query(('v',), lambda: L.load().pg_conv_v_bytes(ctypes.byref(self.g), self.algo, self.ws_arg))

    @staticmethod
    def _extras(part=None, v_keep=None, v_pre=None, u_cache=None, u_valid=False, mul=None):
        if part is None and v_keep is None and v_pre is None and u_cache is None and mul is None:
            return None
        p = lambda t: t.data_ptr() if t is not None else None
        mt, ml, ma = (mul[0].ptr(), mul[0].ld, mul[1]) if mul is not None else (None, 0, 0)
        return L.ConvExtras(p(part), p(v_keep), p(v_pre), p(u_cache), 1 if u_valid else 0, mt, ml, ma)

    def mul_ok(self, small, big, t):
        """small2big(small -> big) can fold `big *= f'(t)` into its epilogue (pg_conv_extras.mul_t): the kernel of this call supports
        it and the tensors satisfy its alignment; t must have big's shape and storage type."""
        if t.bf != big.bf or (t.N, t.H, t.W, t.C) != (big.N, big.H, big.W, big.C) or not self.fits(small, big, t):
            return False
        need = (big, t) if self.Ca == 1 else (small, big, t)       # (the one-channel kernel reads `small` by scalars)
        if not all(v.ptr() % 16 == 0 and v.ld % (8 if v.bf else 4) == 0 for v in need):
            return False
        io = self._io(big, small)
        return bool(self._query(('mul', io), lambda: L.load().pg_conv_mul_ok(ctypes.byref(self.g), self.algo | io,
                                                                              self.ws_arg)))

    def big2small(self, big, P, p_off, bias, b_off, small, act=L.ACT_NONE, part=None, v_keep=None, u_cache=None, u_valid=False):
        """part / v_keep / u_cache: the optional hand-overs of pg_conv_extras (sizes: stats_chunks(0), v_bytes(), u_bytes(0))."""
        assert (big.N, big.H, big.W, big.C) == (self.N, self.Hb, self.Wb, self.Cb), 'big view mismatch'
        assert (small.N, small.H, small.W, small.C) == (self.N, self.Hs, self.Ws, self.Ca), 'small view mismatch'
        wp, wn = self._ws(P.device)
        args = (big.ptr(), big.ld, L.ptr(P, p_off), L.ptr(bias, b_off) if bias is not None else None, small.ptr(), small.ld,
                ctypes.byref(self.g), act, self.algo | self._io(big, small), wp, wn, _stream())
        x = self._extras(part=part, v_keep=v_keep, u_cache=u_cache, u_valid=u_valid)

        def go():
            if x is None:
                L.check(L.load().pg_conv4x4_big2small(*args), 'pg_conv4x4_big2small')
            else:
                L.check(L.load().pg_conv4x4_big2small_x(*args, ctypes.byref(x)), 'pg_conv4x4_big2small_x')
        PROFILER.launch(self, 0, go, self._io(big, small)) if PROFILER is not None else go()

    def small2big(self, small, P, p_off, bias, b_off, big, act=L.ACT_NONE, part=None, u_cache=None, u_valid=False, mul=None):
        """mul = (t, act code): big = conv(...) * f'(t), the activation backward of the layer below folded into the epilogue
        (only where mul_ok(small, big, t))."""
        assert (big.N, big.H, big.W, big.C) == (self.N, self.Hb, self.Wb, self.Cb), 'big view mismatch'
        assert (small.N, small.H, small.W, small.C) == (self.N, self.Hs, self.Ws, self.Ca), 'small view mismatch'
        wp, wn = self._ws(P.device)
        args = (small.ptr(), small.ld, L.ptr(P, p_off), L.ptr(bias, b_off) if bias is not None else None, big.ptr(), big.ld,
                ctypes.byref(self.g), act, self.algo | self._io(big, small), wp, wn, _stream())
        x = self._extras(part=part, u_cache=u_cache, u_valid=u_valid, mul=mul)

        def go():
            if x is None:
                L.check(L.load().pg_conv4x4_small2big(*args), 'pg_conv4x4_small2big')
            else:
                L.check(L.load().pg_conv4x4_small2big_x(*args, ctypes.byref(x)), 'pg_conv4x4_small2big_x')
        PROFILER.launch(self, 1, go, self._io(big, small)) if PROFILER is not None else go()

    def wgrad(self, small, big, dP, p_off, dbias=None, b_off=0, v_pre=None):
        """v_pre: the transformed `big` tensor kept by this layer's big2small(..., v_keep=) (same tensor, v_bytes() > 0)."""
        assert (big.N, big.H, big.W, big.C) == (self.N, self.Hb, self.Wb, self.Cb), 'big view mismatch'
        assert (small.N, small.H, small.W, small.C) == (self.N, self.Hs, self.Ws, self.Ca), 'small view mismatch'
        wp, wn = self._ws(dP.device)
        args = (small.ptr(), small.ld, big.ptr(), big.ld, L.ptr(dP, p_off), L.ptr(dbias, b_off) if dbias is not None else None,
                ctypes.byref(self.g), self.algo | self._io(big, small), wp, wn, _stream())
        x = self._extras(v_pre=v_pre)

        def go():
            if x is None:
                L.check(L.load().pg_conv4x4_wgrad(*args), 'pg_conv4x4_wgrad')
            else:
                L.check(L.load().pg_conv4x4_wgrad_x(*args, ctypes.byref(x)), 'pg_conv4x4_wgrad_x')
        ex = cur_exec(dP.device)
        if ex.allow:
            # on the second stream (see Exec): after everything enqueued so far (dy is ready), with that stream's own workspace
            side = ex.side_stream()
            side.wait_stream(torch.cuda.current_stream())
            ex.keep.append((small.t, big.t, v_pre, dP))
            # what this call adds to the bytes held alive: each backing storage once (the skip buffers are shared by several
            # layers' views), plus the transformed input handed over from the forward pass
            for t in (small.t, big.t, v_pre):
                if t is not None and t.data_ptr() not in ex.keep_ptrs:
                    ex.keep_ptrs.add(t.data_ptr())
                    ex.keep_bytes += t.numel() * t.element_size()
            was, ex.inside = ex.inside, True
            try:
                with torch.cuda.stream(side):
                    wp2, wn2 = self._ws(dP.device)
                    args = args[:8] + (wp2, wn2, _stream())
                    go()
            finally:
                ex.inside = was
            if ex.keep_bytes > KEEP_LIMIT_BYTES:
                # bound what the second stream keeps alive: an intermediate join (the chains re-synchronise once; same results)
                torch.cuda.current_stream().wait_stream(side)
                ex.keep.clear()
                ex.keep_ptrs.clear()
                ex.keep_bytes = 0
            return
        PROFILER.launch(self, 2, go, self._io(big, small)) if PROFILER is not None else go()

    def bwd_big(self, small, big, P, dP, p_off, dsmall, u_cache=None, u_valid=False):
        """Weight gradient (small = x, big = dy) and data gradient (big -> small) of a ConvTranspose2d layer in one call; where
        both run the polyphase Winograd path the transformed dy is computed once and shared (pg_conv4x4_bwd_big).
        u_cache / u_valid: the data gradient's transformed weights, as for big2small."""
        assert (big.N, big.H, big.W, big.C) == (self.N, self.Hb, self.Wb, self.Cb), 'big view mismatch'
        assert (small.N, small.H, small.W, small.C) == (self.N, self.Hs, self.Ws, self.Ca), 'small view mismatch'
        assert (dsmall.N, dsmall.H, dsmall.W, dsmall.C) == (self.N, self.Hs, self.Ws, self.Ca), 'dsmall view mismatch'
        wp, wn = self._ws(P.device)

        def go():
            assert small.bf == big.bf == dsmall.bf, 'bwd_big: one storage type for the three activation tensors'
            args = (small.ptr(), small.ld, big.ptr(), big.ld, L.ptr(P, p_off), L.ptr(dP, p_off), dsmall.ptr(), dsmall.ld,
                    ctypes.byref(self.g), self.algo | self._io(big, small), wp, wn, _stream())
            if u_cache is None:
                L.check(L.load().pg_conv4x4_bwd_big(*args), 'pg_conv4x4_bwd_big')
            else:
                x = self._extras(u_cache=u_cache, u_valid=u_valid)
                L.check(L.load().pg_conv4x4_bwd_big_x(*args, ctypes.byref(x)), 'pg_conv4x4_bwd_big_x')
        PROFILER.launch2(self, (2, 0), go, self._io(big, small)) if PROFILER is not None else go()


def _bf16_tensors_ok(ops, ld_mult=1):
    """Every one of these layers has bf16-tensor kernels for its three ops at this geometry (pg_conv_kernel names them): otherwise
    the caller keeps fp32 activation storage for this extent (e.g. 7 x 7 maps on channel counts the LDS-DMA weight gradient does
    not take: the register-staged kernel's pixel decode needs power-of-two or >= 16-wide maps).  And every tensor those kernels
    would see stays below pg_conv_max_tensor_bytes (bf16 tensors have no kernel beyond it; fp32 ones fall back to the generic
    kernels): ld_mult = the widest buffer an interior tensor is a channel slice of, in multiples of its own channel count (2 for
    the generator's skip-connection buffers)."""
    if _MAX_TENSOR_BYTES is None:
        ConvOp.fits()
    for op in ops:
        if max(op.N * op.Hb * op.Wb * op.Cb, op.N * op.Hs * op.Ws * op.Ca) * ld_mult * 2 >= _MAX_TENSOR_BYTES:
            return False
    return all('bf16' in op.describe(oc, L.IO_MASK)[0] for op in ops for oc in (0, 1, 2))


class UCache(dict):
    """A network's transformed / packed weights for ONE weight version: key -> device buffer (see _ucache).  `log` records how each
    entry was made -- (key, op, opcode, io, p_off, bytes) -- so that the engine can fill the same set in one launch next time."""

    def __init__(self, pool=None):
        super().__init__()
        self.log = []
        self.pool = pool if pool is not None else {}      # key -> device buffer, reused from step to step (see _WeightPrep)

    def buffer(self, key, nb, dev):
        """The device buffer of entry `key`: the pool's (same stream as every reader and writer of it, so reusing it for the next
        weight version needs no synchronisation), allocated on first use -- no allocator traffic per step."""
        buf = self.pool.get(key)
        if buf is None or buf.numel() != nb or buf.device != dev:
            buf = self.pool[key] = torch.empty(nb, dtype=torch.uint8, device=dev)
        return buf


def _ucache(ucache, li, opcode, op, dev, src, dst, p_off=None):
    """(buffer, valid) of the transformed / packed weights of layer li / direction opcode in the caller's per-step cache (a dict that
    lives exactly as long as the weights stay unchanged).  On bf16 tensors one packed bf16 copy serves both directions of a layer (the
    small -> big kernel stages it transposed), so the two directions share an entry."""
    if ucache is None or not CACHE_U or not ConvOp.fits(src, dst):
        return None, False
    big, small = (src, dst) if opcode == 0 else (dst, src)
    io = ConvOp._io(big, small)
    if io:      # bf16 tensors: the LDS-DMA kernels' packed bf16 weights (16-byte-aligned views)
        if not all(v.ptr() % 16 == 0 and v.ld % (8 if v.bf else 4) == 0 for v in (src, dst)):
            return None, False
    elif not ConvOp._aligned(src, dst):
        return None, False
    nb = op.u_bytes(opcode, io)
    if not nb:
        return None, False
    # (the window-staged kernel k_conv_bf16r reads weight fragments straight from global memory and takes a fragment-ordered pack of its
    #  own per direction: its layers keep one entry per direction)
    shared = (io == L.IO_MASK and op.Cb > 8 and not (op.algo & L.TUNE_BF16X_RING)
              and not op.describe(0, io)[0].startswith('k_conv_bf16r') and not op.describe(1, io)[0].startswith('k_conv_bf16r'))
    key = (li, 'w', nb) if shared else (li, opcode, nb)         # nb separates the F(2x2,4x4) / F(3x3,4x4) transforms of the stride-1 layer
    if key in ucache:
        return ucache[key], True
    if isinstance(ucache, UCache):
        ucache[key] = ucache.buffer(key, nb, dev)
        if p_off is not None:
            ucache.log.append((key, op, opcode, io, p_off, nb))
    else:
        ucache[key] = torch.empty(nb, dtype=torch.uint8, device=dev)
    return ucache[key], False


def prefill_ucache(plan, flat, dev, pool=None):
    """A UCache holding every entry of `plan` (the log of an earlier UCache of the same network, input extent and storage mode),
    filled from the CURRENT weights `flat` by ONE pg_conv_prep_batch call: one launch per kernel family instead of one small weight
    transform / pack kernel per layer and direction.  Entries that end up unused cost a few microseconds; entries that are
    missing are made lazily by the conv calls as before."""
    uc = UCache(pool)
    if not plan or not PREP_BATCH:
        return uc
    items = (L.ConvPrepItem * len(plan))()
    for it, (key, op, opcode, io, p_off, nb) in zip(items, plan):
        buf = uc[key] = uc.buffer(key, nb, dev)
        it.g, it.op, it.algo, it.ws_bytes = op.g, opcode, op.algo | io, op.ws_arg
        it.P, it.u = L.ptr(flat, p_off), buf.data_ptr()
    uc.log = list(plan)
    L.check(L.load().pg_conv_prep_batch(len(plan), items, _stream()), 'pg_conv_prep_batch')
    return uc


def _dt(*views):
    """dtype mask of the *_t entry points: bit i = the i-th tensor argument is bf16."""
    m = 0
    for i, v in enumerate(views):
        if v is not None and v.bf:
            m |= 1 << i
    return m


def instnorm_act_fwd(y, out, stats, act, drop_p=0.0, seed=0):
    if y.HW <= 1:
        raise ValueError(f"Expected more than 1 spatial element when training, got input size "
                         f"torch.Size([{y.N}, {y.C}, {y.H}, {y.W}])")
    ws = _workspace(int(L.load().pg_instnorm_workspace_bytes(y.N, y.HW, y.C)), y.t.device)
    L.check(L.load().pg_instnorm_act_fwd_t(y.ptr(), y.ld, out.ptr(), out.ld, stats.data_ptr(), y.N, y.HW, y.C, act, 1e-5,
                                           drop_p, seed & _MASK64, ws.data_ptr(), ws.numel(), _stream(), _dt(y, out)),
            'pg_instnorm_act_fwd')


def conv_instnorm_act(op, opcode, src, flat, p_off, y, out, stats, act, drop_p=0.0, seed=0, v_keep=None, u_cache=None, u_valid=False):
    """Conv2d / ConvTranspose2d (no bias) -> InstanceNorm2d -> activation -> dropout (unet.py:19-30,53-67).  Where the conv's
    kernel can emit the per-sample sums of its output (polyphase Winograd output transform) the InstanceNorm statistics come
    from those partials and the separate statistics pass over y is skipped; otherwise conv, then pg_instnorm_act_fwd."""
    conv = op.big2small if opcode == 0 else op.small2big
    kw = {'v_keep': v_keep} if v_keep is not None else {}
    if u_cache is not None:
        kw.update(u_cache=u_cache, u_valid=u_valid)
    if y.HW <= 1:
        raise ValueError(f"Expected more than 1 spatial element when training, got input size "
                         f"torch.Size([{y.N}, {y.C}, {y.H}, {y.W}])")
    chunks = op.stats_chunks(opcode, src, y) if FUSE_IN_STATS else 0
    if chunks:
        part = torch.empty(y.N * chunks * y.C * 2, dtype=torch.float64, device=y.t.device)
        conv(src, flat, p_off, None, 0, y, part=part, **kw)
        L.check(L.load().pg_instnorm_act_fwd_parts_t(y.ptr(), y.ld, out.ptr(), out.ld, stats.data_ptr(), part.data_ptr(), chunks, y.N,
                                                     y.HW, y.C, act, 1e-5, drop_p, seed & _MASK64, _stream(), _dt(y, out)),
                'pg_instnorm_act_fwd_parts')
    else:
        conv(src, flat, p_off, None, 0, y, **kw)
        instnorm_act_fwd(y, out, stats, act, drop_p, seed)


def _exp_env(name, default='1'):
    """A/B switches for same-device timing: honoured only when PATCHGAN_EXPERIMENT is set (like the C side's pg_exp_env)."""
    return os.environ.get(name, default) if 'PATCHGAN_EXPERIMENT' in os.environ else default


FUSE_IN_STATS = _exp_env('PATCHGAN_FUSE_IN_STATS') != '0'     # InstanceNorm sums from the producing conv
KEEP_V = _exp_env('PATCHGAN_KEEP_V') != '0'                   # transformed input handed from forward to weight gradient
CACHE_U = _exp_env('PATCHGAN_CACHE_U') != '0'                 # per-step cache of transformed / packed discriminator weights
PREP_BATCH = _exp_env('PATCHGAN_PREP_BATCH') != '0'           # one batched weight-preparation launch per network and step
FUSE_ACT_BWD = _exp_env('PATCHGAN_FUSE_ACT_BWD') != '0'       # activation backward in the data-gradient epilogue above it
SEAM8 = _exp_env('PATCHGAN_SEAM8') != '0'       # bf16 storage: image-facing tensors in 8-channel bf16 pixels
WGRAD_SIDE = _exp_env('PATCHGAN_WGRAD_SIDE') != '0'       # weight gradients of a backward pass on a second stream (Exec)
BF16_WGRAD_SIDE = _exp_env('PATCHGAN_BF16_WGRAD_SIDE') != '0'       # bf16 networks' weight gradients on the second stream too (round 5: noise, off; round 6: cfg4 5.41 -> 5.35 ms, cfg2-bf16 3.41 -> 3.33 in three alternating runs each: on)
SHARE_DY_V = _exp_env('PATCHGAN_SHARE_DY_V') != '0'       # ... and the two halves share the transformed dy (A/B switch)
SPLIT_BWD_BIG = _exp_env('PATCHGAN_SPLIT_BWD_BIG') != '0'       # with the second stream: the decoder's fused backward call as its two halves (8.83 -> 8.65 ms at cfg2)


def instnorm_act_bwd(g1, g2, y, stats, dy, act, drop_p=0.0, seed=0):
    ws = _workspace(int(L.load().pg_instnorm_workspace_bytes(y.N, y.HW, y.C)), y.t.device)
    L.check(L.load().pg_instnorm_act_bwd_t(g1.ptr(), g1.ld, g2.ptr() if g2 is not None else None,
                                           g2.ld if g2 is not None else 0, y.ptr(), y.ld, stats.data_ptr(), dy.ptr(), dy.ld,
                                           y.N, y.HW, y.C, act, drop_p, seed & _MASK64, ws.data_ptr(), ws.numel(), _stream(),
                                           _dt(g1, g2, y, dy)), 'pg_instnorm_act_bwd')


def act_bwd(g1, g2, a, dy, act):
    L.check(L.load().pg_act_bwd_t(g1.ptr(), g1.ld, g2.ptr() if g2 is not None else None, g2.ld if g2 is not None else 0,
                                  a.ptr() if a is not None else None, a.ld if a is not None else 0, dy.ptr(), dy.ld,
                                  dy.npix, dy.C, act, 0.0, 0, _stream(), _dt(g1, g2, a, dy)), 'pg_act_bwd')


def softmax_fwd(y, out):
    L.check(L.load().pg_softmax_fwd(y.ptr(), y.ld, out.ptr(), out.ld, y.npix, y.C, _stream()), 'pg_softmax_fwd')


def softmax_bwd(g1, g2, out, dy):
    L.check(L.load().pg_softmax_bwd(g1.ptr(), g1.ld, g2.ptr() if g2 is not None else None,
                                    g2.ld if g2 is not None else 0, out.ptr(), out.ld, dy.ptr(), dy.ld, dy.npix, dy.C,
                                    _stream()), 'pg_softmax_bwd')


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (reference trainer.py:169-172); bias corrections in Python doubles like torch."""
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    L.check(L.load().pg_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2,
                                  eps, bc1, math.sqrt(bc2), _stream()), 'pg_adam_step')


def adam_scalars(step, lr, beta1=0.9, beta2=0.999):
    """(lr / bc1, sqrt(bc2)) as the float32 values pg_adam_step forms from its arguments: what pg_adam_step_dev reads from device memory."""
    bc1 = np.float32(1.0 - beta1 ** step)
    return np.float32(lr) / bc1, np.float32(math.sqrt(1.0 - beta2 ** step))


def adam_step_dev(p, g, m, v, scalars, beta1=0.9, beta2=0.999, eps=1e-8):
    """adam_step with lr / bc1 and sqrt(bc2) taken from the device tensor `scalars` (2 floats): a launch without step-dependent arguments."""
    L.check(L.load().pg_adam_step_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), beta1, beta2, eps,
                                      scalars.data_ptr(), _stream()), 'pg_adam_step_dev')


def tiles_gather(image, size, overlap):
    """reference infer.py:14-35 (n_crop): device image [C, H, W] -> View of the ny*nx overlapping size x size tiles in the
    NHWC batch layout the generator kernels read."""
    C, H, W = image.shape
    eff = int(overlap * size)
    lib = L.load()
    ny, nx = lib.pg_tiles_count(H, size, eff), lib.pg_tiles_count(W, size, eff)
    if ny <= 0 or nx <= 0:
        raise ValueError(f"cannot cut {size}x{size} tiles with overlap {overlap} from a {H}x{W} image")
    image = image.to(dtype=torch.float32).contiguous()
    tiles = View.alloc(ny * nx, size, size, C, image.device)
    L.check(lib.pg_tiles_gather(image.data_ptr(), C, H, W, size, eff, tiles.ptr(), tiles.ld, _stream()), 'pg_tiles_gather')
    return tiles


def tiles_blend(tiles, image_size, threshold, overlap):
    """reference infer.py:38-68 (build_mask): View of predicted tiles -> overlap-averaged mask on the device: float64
    [H, W] for one class, int64 class index [H, W] for several (same values / dtypes as the reference returns)."""
    H, W = image_size
    size = tiles.H
    eff = int(overlap * size)
    lib = L.load()
    if lib.pg_tiles_count(H, size, eff) * lib.pg_tiles_count(W, size, eff) != tiles.N or tiles.W != size:
        raise ValueError(f"{tiles.N} tiles of {tiles.H}x{tiles.W} do not tile a {H}x{W} image with overlap {overlap}")
    dev = tiles.t.device
    if tiles.C > 1:
        out = torch.empty(H, W, dtype=torch.int64, device=dev)
        mask_p, arg_p = None, out.data_ptr()
    else:
        out = torch.empty(H, W, dtype=torch.float64, device=dev)
        mask_p, arg_p = out.data_ptr(), None
    L.check(lib.pg_tiles_blend(tiles.ptr(), tiles.ld, tiles.C, size, eff, H, W, float(threshold), mask_p, arg_p, _stream()),
            'pg_tiles_blend')
    return out


_GOLD = 0x9E3779B97F4A7C15


def _shift_seed(seed, elem_offset):
    """Seed whose dropout stream is `seed`'s shifted by elem_offset elements: pg_dropout_keep hashes
    seed + GOLD * (e + 1), so the mask of local element e under the shifted seed is the mask of global element
    e + elem_offset under `seed` (data parallelism: rank r's samples sit at r*N.. of the global batch)."""
    return (seed + _GOLD * elem_offset) & _MASK64


def _mix_seed(base, *vals):
    h = (base * 0x9E3779B97F4A7C15 + 0x1234567) & _MASK64
    for v in vals:
        h ^= (v + 0x9E3779B97F4A7C15 + ((h << 6) & _MASK64) + (h >> 2)) & _MASK64
        h = (h * 0xBF58476D1CE4E5B9) & _MASK64
    return h


# ------------------------------------------------------------------------------------------------
# parameter layout
# ------------------------------------------------------------------------------------------------


class LayerSpec:
    """One conv layer of a network: state_dict key, torch weight shape [a, b, 4, 4], offsets into the flat buffer."""
    __slots__ = ('key', 'a', 'b', 'stride', 'transposed', 'p_off', 'bias_key', 'b_off', 'act', 'norm', 'dropout')

    def __init__(self, key, a, b, stride, transposed, act, norm, dropout=False, bias_key=None):
        self.key, self.a, self.b, self.stride, self.transposed = key, a, b, stride, transposed
        self.act, self.norm, self.dropout, self.bias_key = act, norm, dropout, bias_key
        self.p_off = self.b_off = -1

    @property
    def cout(self):
        return self.b if self.transposed else self.a

    @property
    def cin(self):
        return self.a if self.transposed else self.b


def assign_offsets(layers):
    """Lay the packed blocks out back to back (16-byte aligned); returns the flat length in floats."""
    off = 0
    for l in layers:
        l.p_off = off
        off += 16 * l.a * l.b
        if l.bias_key is not None:
            l.b_off = off
            off += (l.cout + 3) // 4 * 4
    return off


def torch_views(flat, layers):
    """{state_dict key: strided view of `flat` with the reference's shape} (weights [a,b,4,4], biases [cout])."""
    out = {}
    for l in layers:
        ab = l.a * l.b
        out[l.key] = flat.as_strided((l.a, l.b, 4, 4), (l.b, 1, 4 * ab, ab), l.p_off)
        if l.bias_key is not None:
            out[l.bias_key] = flat.as_strided((l.cout,), (1,), l.b_off)
    return out


def default_init_(flat, layers, generator=None):
    """torch's default Conv2d / ConvTranspose2d init (the reference's ``weights_init`` is a no-op,
    trainer.py:327-343): kaiming_uniform_(a=sqrt(5)) = U(+-1/sqrt(fan_in)), fan_in = size(1)*16; bias
    U(+-1/sqrt(fan_in)).  Drawn in state_dict order with the CPU generator like nn.Module construction would."""
    views = torch_views(flat, layers)
    with torch.no_grad():
        for l in layers:
            bound = 1.0 / math.sqrt(l.b * 16)
            w = torch.empty(l.a, l.b, 4, 4).uniform_(-bound, bound, generator=generator)
            views[l.key].copy_(w)
            if l.bias_key is not None:
                bb = torch.empty(l.cout).uniform_(-bound, bound, generator=generator)
                views[l.bias_key].copy_(bb)


# ------------------------------------------------------------------------------------------------
# generator
# ------------------------------------------------------------------------------------------------


def unet_layers(input_nc, output_nc, nf, activation, final_act, use_dropout):
    """Layer plan of reference UNet.__init__ (unet.py:84-107)."""
    if activation not in ('tanh', 'relu', 'leakyrelu'):
        raise ValueError(f"activation must be one of tanh|relu|leakyrelu, got {activation!r}")
    if final_act not in ('tanh', 'relu', 'leakyrelu', 'sigmoid', 'softmax'):
        raise ValueError(f"final_act must be one of tanh|relu|leakyrelu|sigmoid|softmax, got {final_act!r}")
    filts = [nf, nf * 2, nf * 4, nf * 8, nf * 8, nf * 8, nf * 8]
    enc, dec = [], []
    prev = input_nc
    for i, f in enumerate(filts):
        enc.append(LayerSpec(f'encoder.{i}.model.DownConv{i}.weight', f, prev, 2, False, activation, True, use_dropout))
        prev = f
    for i, f in enumerate(filts[:-1][::-1]):
        cin = prev if i == 0 else prev * 2
        dec.append(LayerSpec(f'decoder.{i}.model.UpConv{i}.weight', cin, f, 2, True, activation, i != 0,
                             use_dropout and i != 0))
        prev = f
    dec.append(LayerSpec('decoder.6.model.UpConv6.weight', nf * 2, output_nc, 2, True, final_act, False, False))
    return enc, dec


class GenContext:
    """Activations saved by one generator forward (what autograd would have kept)."""
    pass


class _WeightPrep:
    """Per-step weight preparation of a network (mixed into the two engines).  ucache_begin() opens the cache for one weight version:
    from the second step of a given kind on it comes back already filled by one batched launch (prefill_ucache) with what the previous
    such step used; ucache_end() remembers what this step used.

    Memory: the transformed / packed weights depend on the layer, the direction and the transform (all three are in the entry key,
    which carries the byte count), NOT on the input extent -- so the device buffers live in ONE pool per network keyed by entry,
    shared by every extent and mode (a ragged last batch or a second image size adds no device memory).  The per-(extent, mode)
    fill plans are small host lists, kept for the MAX_PLANS most recently used kinds of step.  clear_weight_caches() (called by
    set_precision / set_tuning) drops both."""

    MAX_PLANS = 8

    def ucache_begin(self, flat, N, H, W, tag=None):
        key = (N, H, W, tag, self.algo, bool(self.act_bf))
        pool = self.__dict__.setdefault('_upool', {})
        plans = self.__dict__.setdefault('_uplan', {})
        plan = plans.pop(key, None)
        if plan is not None:
            plans[key] = plan          # most recently used last
        uc = prefill_ucache(plan, flat, flat.device, pool)
        uc.plan_key = key
        return uc

    def ucache_end(self, uc):
        if isinstance(uc, UCache) and getattr(uc, 'plan_key', None) is not None:
            plans = self.__dict__.setdefault('_uplan', {})
            plans.pop(uc.plan_key, None)
            plans[uc.plan_key] = list(uc.log)
            while len(plans) > self.MAX_PLANS:
                plans.pop(next(iter(plans)))
            # buffers no remembered plan refers to any more (entries of evicted plans) go back to the allocator
            pool = self.__dict__.get('_upool', {})
            live = {e[0] for p in plans.values() for e in p} | set(uc.keys())
            for k in [k for k in pool if k not in live]:
                del pool[k]

    def clear_weight_caches(self):
        self.__dict__.pop('_upool', None)
        self.__dict__.pop('_uplan', None)


class GeneratorEngine(_WeightPrep):
    _LD_MULT = 2          # interior tensors are channel slices of the skip-connection buffers cat_i (twice their channels)

    def __init__(self, input_nc, output_nc, nf, activation, final_act, use_dropout, algo=None):
        self.input_nc, self.output_nc, self.nf = input_nc, output_nc, nf
        self.activation, self.final_act, self.use_dropout = activation, final_act, use_dropout
        self.enc, self.dec = unet_layers(input_nc, output_nc, nf, activation, final_act, use_dropout)
        self.layers = self.enc + self.dec
        self.nparams = assign_offsets(self.layers)
        self.algo = DEFAULT_ALGO if algo is None else algo
        self.act_bf = False          # bf16 activation storage (set_precision('bf16'); needs nf % 4 == 0 and nf >= 32)
        self._ops = {}

    def bf16_storage_ok(self):
        """Every interior tensor feeds / comes from a fast bf16 conv kernel: channel counts multiples of 4 and >= 32."""
        return self.nf % 4 == 0 and self.nf >= 32

    def _storage_ok(self, key, interior_ops):
        if not hasattr(self, '_sok'):
            self._sok = {}
        if key not in self._sok:
            self._sok[key] = _bf16_tensors_ok(interior_ops, self._LD_MULT)
        return self._sok[key]

    def ops(self, N, H, W):
        """ConvOps for input extent (N, H, W); cached."""
        key = (N, H, W)
        if key not in self._ops:
            enc_ops, dec_ops = [], []
            h, w = H, W
            for l in self.enc:
                op = ConvOp(N, h, w, l.a, l.b, 2, self.algo)
                enc_ops.append(op)
                h, w = op.Hs, op.Ws
                if h * w <= 1:     # torch raises from this block's InstanceNorm before the next conv is reached
                    raise ValueError(f"Expected more than 1 spatial element when training, got input size "
                                     f"torch.Size([{N}, {l.a}, {h}, {w}])")
            sizes = [(op.Hs, op.Ws) for op in enc_ops]   # enc i output extent
            for i, l in enumerate(self.dec):
                # convT: small = input (h, w), big = output (2h, 2w)
                op = ConvOp(N, 2 * h, 2 * w, l.a, l.b, 2, self.algo)
                assert (op.Hs, op.Ws) == (h, w)
                dec_ops.append(op)
                h, w = 2 * h, 2 * w
                if i < 6:
                    sh, sw = sizes[5 - i]
                    if (sh, sw) != (h, w):
                        raise RuntimeError(f"Sizes of tensors must match except in dimension 1. Expected size {h} but got "
                                           f"size {sh} for the skip connection of decoder block {i + 1} (input {H}x{W}: "
                                           "H and W must be multiples of 128)")
            self._ops[key] = (enc_ops, dec_ops)
        return self._ops[key]

    def forward(self, flat, xin, gen_out, train, seed=0, sample0=0, keep_v=False, ucache=None):
        """xin: View [N,H,W,input_nc]; gen_out: View [N,H,W,output_nc] to receive final_act(dec6).
        train selects dropout (InstanceNorm always uses instance statistics, unet.py:77).  sample0 = index of this
        batch's first sample in the global batch (data parallelism: the dropout masks are those of the global batch).
        keep_v: a backward pass follows -- encoder layers on the polyphase Winograd path keep their transformed input for the
        weight gradient (pg_conv_extras.v_keep / v_pre) instead of transforming it again.
        ucache: a dict the caller keeps for as long as `flat` is unchanged (Trainer.batch: forward + backward of one step): on bf16
        tensors a layer's packed bf16 weights are built once and serve both its forward and its data gradient."""
        N, H, W = xin.N, xin.H, xin.W
        dev = flat.device
        enc_ops, dec_ops = self.ops(N, H, W)
        F = [l.a for l in self.enc]
        c = GenContext()
        c.N, c.H, c.W, c.xin, c.gen_out, c.seed, c.train, c.sample0 = N, H, W, xin, gen_out, seed, train, sample0
        # cat_i (i = 1..6): input of decoder i = [dec_{i-1} out | enc_{6-i} out]
        # bf16 activation storage: every interior activation (conv outputs, normalised outputs, their gradients) is a bf16 tensor;
        # the image-facing tensors (x, enc0's conv output, the generator output and its gradient) stay fp32 and the InstanceNorm /
        # activation kernels next to them read one type and write the other
        bf = c.bf = bool(self.act_bf) and self._storage_ok((N, H, W), enc_ops[1:] + dec_ops[:6])
        # ... unless they fit 8-channel pixels: then x and dL/d(output) are padded to that bf16 layout (one small pass each) and the
        # first / last layer run on the bf16 kernels as well (SEAM8)
        # (nf % 64: the row GEMMs next to the images -- the head's forward over 2 * nf channels, the data gradient onto x over nf --
        # contract in 64-wide chunks)
        seam8 = c.seam8 = (bf and SEAM8 and self.nf % 64 == 0 and self.input_nc <= 8 and self.output_nc <= 8
                           and enc_ops[0].describe(0, L.IO_MASK)[0].startswith('k_conv_bf16x'))      # (not with PG_TUNE_BF16X_OFF)
        c.xin8 = xin.padded8() if seam8 else None
        c.cat = [None] * 7
        for i in range(1, 7):
            op = dec_ops[i]
            c.cat[i] = View.alloc(N, op.Hs, op.Ws, self.dec[i].a, dev, bf=bf)
        c.hidden = View.alloc(N, enc_ops[6].Hs, enc_ops[6].Ws, F[6], dev, bf=bf)
        c.y, c.stats, c.enc_out, c.v = [], [], [], []
        act = L.ACT_CODES[self.activation]
        src = c.xin8 if seam8 else xin
        for i, (l, op) in enumerate(zip(self.enc, enc_ops)):
            y = View.alloc(N, op.Hs, op.Ws, l.a, dev, bf=bf and (i > 0 or seam8))
            if i < 6:
                cat = c.cat[6 - i]
                out = cat.channels(cat.C - l.a, l.a)
            else:
                out = c.hidden
            stats = torch.empty(N * l.a * 2, dtype=torch.float32, device=dev)
            drop = 0.2 if (train and l.dropout) else 0.0
            vb = op.v_bytes() if (keep_v and KEEP_V and ConvOp._aligned(src, y) and ConvOp.fits(src, y)) else 0
            vk = torch.empty(vb, dtype=torch.uint8, device=dev) if vb else None
            u, uv = _ucache(ucache, ('e', i), 0, op, dev, src, y, l.p_off)
            conv_instnorm_act(op, 0, src, flat, l.p_off, y, out, stats, act, drop, _shift_seed(_mix_seed(seed, 1, i), sample0 * y.HW * y.C),
                              v_keep=vk, u_cache=u, u_valid=uv)
            c.v.append(vk)
            c.y.append(y)
            c.stats.append(stats)
            c.enc_out.append(out)
            src = out
        c.yd, c.statsd = [None] * 7, [None] * 7
        src = c.hidden
        for i, (l, op) in enumerate(zip(self.dec, dec_ops)):
            if i == 6:
                # the 1..4-channel head: taps folded into N (row GEMM + col2im); fp32 kernels on an fp32 copy of its input, or the
                # bf16 row GEMM straight from the bf16 tensor
                src = c.cat6_f32 = src if seam8 else src.converted(False)
                if self.final_act == 'softmax':
                    c.gen_raw = View.alloc(N, op.Hb, op.Wb, l.b, dev)
                    op.small2big(src, flat, l.p_off, None, 0, c.gen_raw)
                    softmax_fwd(c.gen_raw, gen_out)
                else:
                    op.small2big(src, flat, l.p_off, None, 0, gen_out, L.ACT_CODES[self.final_act])
                break
            cat = c.cat[i + 1]
            out = cat.channels(0, l.b)
            if l.norm:
                yd = View.alloc(N, op.Hb, op.Wb, l.b, dev, bf=bf)
                stats = torch.empty(N * l.b * 2, dtype=torch.float32, device=dev)
                drop = 0.2 if (train and l.dropout) else 0.0
                u, uv = _ucache(ucache, ('d', i), 1, op, dev, src, yd, l.p_off)
                conv_instnorm_act(op, 1, src, flat, l.p_off, yd, out, stats, act, drop,
                                  _shift_seed(_mix_seed(seed, 2, i), sample0 * yd.HW * yd.C), u_cache=u, u_valid=uv)
                c.yd[i], c.statsd[i] = yd, stats
            else:
                op.small2big(src, flat, l.p_off, None, 0, out, act)
            src = cat
        return c

    def backward(self, flat, gflat, c, g1, g2=None, need_dx=False, on_ready=None, ucache=None, defer_join=False):
        """g1 (+ g2): Views of dL/d(gen_out).  Writes every weight gradient into `gflat` (packed layout);
        returns dL/dx as a View if need_dx.  on_ready(lo, hi) is called after each layer's weight gradient has been
        enqueued with that layer's range of the flat buffer (last layer first): the hook data parallelism uses to
        start all-reducing finished buckets under the rest of the backward pass."""
        def done(l):
            if on_ready is not None:
                on_ready(l.p_off, l.p_off + 16 * l.a * l.b)
        # defer_join: the caller joins the second stream itself (side_join()) before it reads the weight gradients -- the trainer puts
        # the discriminator's forward pass, which needs none of them, in between
        _side_begin(not self.act_bf or BF16_WGRAD_SIDE)       # (bf16: the two chains contend for the L1 path: 5.87 -> 5.94 ms at cfg4)
        ok = False
        try:
            r = self._backward(flat, gflat, c, g1, g2, need_dx, done, ucache)
            ok = True
            return r
        finally:
            if not (ok and defer_join):
                _side_join()
            else:
                cur_exec().allow = False

    def _backward(self, flat, gflat, c, g1, g2, need_dx, done, ucache):
        N, dev = c.N, flat.device
        enc_ops, dec_ops = self.ops(c.N, c.H, c.W)
        act = L.ACT_CODES[self.activation]
        # ---- decoder, last to first
        l, op = self.dec[6], dec_ops[6]
        dy = View.alloc(N, op.Hb, op.Wb, l.b, dev)
        if self.final_act == 'softmax':
            softmax_bwd(g1, g2, c.gen_out, dy)
        else:
            act_bwd(g1, g2, c.gen_out, dy, L.ACT_CODES[self.final_act])
        bf = c.bf
        if c.seam8:
            dy = dy.padded8()
        dcat = View.alloc(N, op.Hs, op.Ws, l.a, dev, bf=c.seam8)  # fp32 with the head's fp32 kernels; the blocks below read it as is
        op.bwd_big(c.cat6_f32, dy, flat, gflat, l.p_off, dcat)    # weight gradient + data gradient of the ConvTranspose2d
        done(l)
        dskip = [None] * 7   # dskip[j]: gradient wrt enc_j output arriving through the skip connection
        for i in range(5, -1, -1):
            l, op = self.dec[i], dec_ops[i]
            g = dcat.channels(0, l.b)
            dskip[5 - i] = dcat.channels(l.b, dcat.C - l.b)
            dy = View.alloc(N, op.Hb, op.Wb, l.b, dev, bf=bf)
            if l.norm:
                drop = 0.2 if (c.train and l.dropout) else 0.0
                instnorm_act_bwd(g, None, c.yd[i], c.statsd[i], dy, act, drop,
                                 _shift_seed(_mix_seed(c.seed, 2, i), c.sample0 * dy.HW * dy.C))
            else:
                act_bwd(g, None, c.cat[i + 1].channels(0, l.b), dy, act)
            src = c.hidden if i == 0 else c.cat[i]
            dsrc = View.alloc(N, op.Hs, op.Ws, l.a, dev, bf=bf)
            u, uv = _ucache(ucache, ('d', i), 0, op, dev, dy, dsrc, l.p_off)
            if (u is not None and bf) or (SPLIT_BWD_BIG and cur_exec().allow):      # bf16 tensors: the two halves as two calls, the data gradient on the forward's packed weights
                vb = op.v_bytes() if (SHARE_DY_V and not bf and KEEP_V and ConvOp._aligned(dy, dsrc, src) and ConvOp.fits(dy, dsrc, src)) else 0
                if vb:
                    # both halves start from the polyphase transform of dy: the data gradient keeps it (v_keep), the weight gradient --
                    # on the second stream, behind this layer's data gradient instead of beside it -- takes it over (v_pre)
                    vk = torch.empty(vb, dtype=torch.uint8, device=dev)
                    op.big2small(dy, flat, l.p_off, None, 0, dsrc, u_cache=u, u_valid=uv, v_keep=vk)
                    op.wgrad(src, dy, gflat, l.p_off, v_pre=vk)
                else:
                    op.wgrad(src, dy, gflat, l.p_off)
                    op.big2small(dy, flat, l.p_off, None, 0, dsrc, u_cache=u, u_valid=uv)
            else:
                op.bwd_big(src, dy, flat, gflat, l.p_off, dsrc, u_cache=u, u_valid=uv)
            done(l)
            dcat = dsrc
        # ---- encoder, last to first; dcat is now dL/d(hidden)
        g_main = dcat
        dx = None
        for j in range(6, -1, -1):
            l, op = self.enc[j], enc_ops[j]
            dy = View.alloc(N, op.Hs, op.Ws, l.a, dev, bf=bf and (j > 0 or c.seam8))
            drop = 0.2 if (c.train and l.dropout) else 0.0
            instnorm_act_bwd(g_main, dskip[j] if j < 6 else None, c.y[j], c.stats[j], dy, act, drop,
                             _shift_seed(_mix_seed(c.seed, 1, j), c.sample0 * dy.HW * dy.C))
            src = (c.xin8 if c.seam8 else c.xin) if j == 0 else c.enc_out[j - 1]
            op.wgrad(dy, src, gflat, l.p_off, v_pre=c.v[j] if ConvOp._aligned(dy, src) else None)
            done(l)
            if j > 0 or need_dx:
                dsrc = View.alloc(N, op.Hb, op.Wb, l.b, dev, bf=bf and j > 0)
                u, uv = _ucache(ucache, ('e', j), 1, op, dev, dy, dsrc, l.p_off)
                op.small2big(dy, flat, l.p_off, None, 0, dsrc, u_cache=u, u_valid=uv)
                if j == 0:
                    dx = dsrc
                g_main = dsrc
        return dx


# ------------------------------------------------------------------------------------------------
# discriminator
# ------------------------------------------------------------------------------------------------


def disc_layers(input_nc, ndf, n_layers, norm):
    """Layer plan of reference Discriminator.__init__ (disc.py:19-46); keys are nn.Sequential indices."""
    layers = []
    idx = 0
    layers.append(LayerSpec(f'model.{idx}.weight', ndf, input_nc, 2, False, 'leakyrelu', False, bias_key=f'model.{idx}.bias'))
    idx += 2
    mult = 1
    for n in range(1, n_layers):
        prev, mult = mult, min(2 ** n, 8)
        layers.append(LayerSpec(f'model.{idx}.weight', ndf * mult, ndf * prev, 2, False, 'tanh', norm))
        idx += 3 if norm else 2
    prev, mult = mult, min(2 ** n_layers, 8)
    layers.append(LayerSpec(f'model.{idx}.weight', ndf * mult, ndf * prev, 1, False, 'tanh', norm))
    idx += 3 if norm else 2
    layers.append(LayerSpec(f'model.{idx}.weight', 1, ndf * mult, 1, False, 'sigmoid', False, bias_key=f'model.{idx}.bias'))
    return layers


class DiscContext:
    pass


class DiscriminatorEngine(_WeightPrep):
    _LD_MULT = 1

    def __init__(self, input_nc, ndf, n_layers, norm, algo=None):
        self.input_nc, self.ndf, self.n_layers, self.norm = input_nc, ndf, n_layers, norm
        self.layers = disc_layers(input_nc, ndf, n_layers, norm)
        self.nparams = assign_offsets(self.layers)
        self.algo = DEFAULT_ALGO if algo is None else algo
        self.act_bf = False          # bf16 activation storage (set_precision('bf16'); needs ndf % 4 == 0 and ndf >= 32)
        self._ops = {}

    def bf16_storage_ok(self):
        return self.ndf % 4 == 0 and self.ndf >= 32

    def _storage_ok(self, key, interior_ops):
        if not hasattr(self, '_sok'):
            self._sok = {}
        if key not in self._sok:
            self._sok[key] = _bf16_tensors_ok(interior_ops, self._LD_MULT)
        return self._sok[key]

    def ops(self, N, H, W):
        key = (N, H, W)
        if key not in self._ops:
            ops = []
            h, w = H, W
            for l in self.layers:
                op = ConvOp(N, h, w, l.a, l.b, l.stride, self.algo)
                ops.append(op)
                h, w = op.Hs, op.Ws
            self._ops[key] = ops
        return self._ops[key]

    def out_shape(self, N, H, W):
        op = self.ops(N, H, W)[-1]
        return (N, 1, op.Hs, op.Ws)

    def forward(self, flat, din, ucache=None, keep_v=False):
        """din: View [N,H,W,input_nc].  Returns a context; ctx.out is the sigmoid patch map View [N,h,w,1].
        ucache: a dict owned by the caller that lives exactly as long as the weights in `flat` stay unchanged (Trainer.batch: one
        step) -- the Winograd-transformed weights are computed once per (layer, direction) and reused by every pass that
        shares the dict.  keep_v: a backward pass with weight gradients follows (see GeneratorEngine.forward)."""
        ops = self.ops(din.N, din.H, din.W)
        dev = flat.device
        c = DiscContext()
        c.din, c.N, c.H, c.W = din, din.N, din.H, din.W
        c.t, c.stats, c.a, c.v, c.src = [], [], [], [], []
        # bf16 activation storage: the tensors between the first and the last layer are bf16; the input (x | mask), the first
        # layer's conv output (4-channel input: generic kernel) and the 1-channel head stay fp32
        last = len(self.layers) - 1
        bf = c.bf = bool(self.act_bf) and self._storage_ok((din.N, din.H, din.W), ops[1:last])
        # the input in 8-channel bf16 pixels where it fits: the first layer then runs on the bf16 kernels too (GeneratorEngine.forward)
        # (ndf % 64: the data gradient onto x | mask is a row GEMM contracting over ndf channels in 64-wide chunks)
        seam8 = c.seam8 = (bf and SEAM8 and self.ndf % 64 == 0 and self.input_nc <= 8 and last > 0
                           and ops[0].describe(0, L.IO_MASK)[0].startswith('k_conv_bf16x'))

        src = din.padded8() if seam8 else din
        for li, (l, op) in enumerate(zip(self.layers, ops)):
            if li == last:
                src = src.converted(False)
            c.src.append(src)
            t = View.alloc(din.N, op.Hs, op.Ws, l.a, dev, bf=bf and (0 < li < last or (li == 0 and seam8)))
            bias = flat if l.bias_key is not None else None
            u, uv = _ucache(ucache, li, 0, op, dev, src, t, l.p_off)
            vb = op.v_bytes() if (keep_v and KEEP_V and ConvOp._aligned(src, t) and ConvOp.fits(src, t)) else 0
            vk = torch.empty(vb, dtype=torch.uint8, device=dev) if vb else None
            c.v.append(vk)
            op.big2small(src, flat, l.p_off, bias, l.b_off, t, L.ACT_CODES[l.act], v_keep=vk, u_cache=u, u_valid=uv)   # conv + bias + act fused
            if l.norm:                                                              # disc.py:31-32: Conv -> Tanh -> IN
                a = View.alloc(din.N, op.Hs, op.Ws, l.a, dev, bf=bf and li < last)
                stats = torch.empty(din.N * l.a * 2, dtype=torch.float32, device=dev)
                instnorm_act_fwd(t, a, stats, L.ACT_NONE)
            else:
                if bf and li == 0 and not seam8:
                    t = t.converted(True)          # the next layer's kernel reads bf16
                a, stats = t, None
            c.t.append(t)
            c.stats.append(stats)
            c.a.append(a)
            src = a
        c.out = src
        return c

    def backward(self, flat, gflat, c, gout, need_wgrad=True, need_dx=False, ucache=None):
        """gout: View of dL/d(out).  need_wgrad=False skips the weight gradients (the generator step's pass
        through D, whose D-gradients the reference zeroes at trainer.py:93-94).  ucache: as in forward."""
        _side_begin(need_wgrad and (not self.act_bf or BF16_WGRAD_SIDE))
        try:
            return self._backward(flat, gflat, c, gout, need_wgrad, need_dx, ucache)
        finally:
            _side_join()

    def _backward(self, flat, gflat, c, gout, need_wgrad, need_dx, ucache):
        ops = self.ops(c.N, c.H, c.W)
        dev = flat.device
        g = gout
        dx = None
        bf, last = c.bf, len(self.layers) - 1
        inner_of = lambda i: bf and (0 < i < last or (i == 0 and c.seam8))
        fused = None        # dy of this layer, already produced by the data-gradient kernel of the layer above
        for li in range(last, -1, -1):
            l, op = self.layers[li], ops[li]
            inner = inner_of(li)
            if fused is not None:
                dy, fused = fused, None
            else:
                if l.norm:
                    dt = View.alloc(c.N, op.Hs, op.Ws, l.a, dev, bf=inner)
                    instnorm_act_bwd(g, None, c.t[li], c.stats[li], dt, L.ACT_NONE)
                    g = dt
                dy = View.alloc(c.N, op.Hs, op.Ws, l.a, dev, bf=inner)
                act_bwd(g, None, c.t[li], dy, L.ACT_CODES[l.act])
            src = c.src[li]
            if need_wgrad:
                op.wgrad(dy, src, gflat, l.p_off, gflat if l.bias_key is not None else None, l.b_off,
                         v_pre=c.v[li] if ConvOp._aligned(dy, src) else None)
            if li > 0 or need_dx:
                # the activation backward of the layer below (no InstanceNorm in between) in this kernel's epilogue: the tensor it
                # writes is then that layer's dy (and has dy's storage type)
                below = self.layers[li - 1] if li > 0 else None
                dsrc = None
                if FUSE_ACT_BWD and below is not None and not below.norm:
                    cand = View.alloc(c.N, op.Hb, op.Wb, l.b, dev, bf=inner_of(li - 1))
                    if op.mul_ok(dy, cand, c.t[li - 1]):
                        dsrc = fused = cand
                if dsrc is None:
                    dsrc = View.alloc(c.N, op.Hb, op.Wb, l.b, dev, bf=bf and 0 < li < last)
                u, uv = _ucache(ucache, li, 1, op, dev, dy, dsrc, l.p_off)
                if fused is not None:
                    op.small2big(dy, flat, l.p_off, None, 0, dsrc, u_cache=u, u_valid=uv, mul=(c.t[li - 1], L.ACT_CODES[below.act]))
                else:
                    op.small2big(dy, flat, l.p_off, None, 0, dsrc, u_cache=u, u_valid=uv)
                g = dsrc
                if li == 0:
                    dx = dsrc
        return dx


# ------------------------------------------------------------------------------------------------
# losses on the device
# ------------------------------------------------------------------------------------------------


class LossBuffers:
    """Scratch for one loss evaluation: S f64 (the per-(sample, channel) reduction terms: COMBINED [N*C*5] sums on the staged path,
    the nsplit partial slabs on the one-launch path, which combines them inside pg_loss_value_grad), sums [2] f64, coef [N*C*2] f32.
    Kept alive by the caller until the launches that use it are enqueued; nothing reads it afterwards."""

    def __init__(self, N, C, device, HW=1):
        self.S = torch.empty(int(L.load().pg_loss_reduce_doubles(N, HW, C)), dtype=torch.float64, device=device)
        self.sums = torch.empty(2, dtype=torch.float64, device=device)      # (written by pg_loss_prepare where it is read at all)
        self.coef = torch.empty(N * C * 2, dtype=torch.float32, device=device)


class PendingLoss:
    """A loss evaluation whose batch-global reduction terms are still being summed across ranks."""
    __slots__ = ('buf', 'wait', 'args', 'nsplit', 'sums')


def _fused_loss_max_nc():
    """pg_loss_value_grad's table (N * C entries) lives in LDS: the library says how many fit (pg_loss_fused_max_nc)."""
    global _FUSED_LOSS_MAX_NC
    if _FUSED_LOSS_MAX_NC is None:
        _FUSED_LOSS_MAX_NC = int(L.load().pg_loss_fused_max_nc())
    return _FUSED_LOSS_MAX_NC


_FUSED_LOSS_MAX_NC = None


def loss_begin(p, y, tconst, beta=0.75, allreduce=None, need_sums=True):
    """Phase 1 of a loss term over prediction View p against target View y (or the constant tconst): the per-sample
    reductions and the two batch-global terms.  `allreduce(tensor)` starts their SUM across ranks under data parallelism
    and returns a wait() callable; anything enqueued between loss_begin and loss_finish overlaps that exchange.
    In one process there is nothing to exchange between the phases: phase 1 is then ONE launch (pg_loss_reduce_parts) and phase 2
    one more (pg_loss_value_grad adds the partial slabs, forms the batch-global terms, the value and the gradient)."""
    lib = L.load()
    buf = LossBuffers(p.N, p.C, p.t.device, p.HW)
    st = _stream()
    h = PendingLoss()
    h.buf, h.args, h.wait, h.sums = buf, (p, y, tconst, beta), None, None
    yp, yl = (y.ptr(), y.ld) if y is not None else (None, 0)
    exchange = allreduce is not None and need_sums
    if not exchange and p.N * p.C <= _fused_loss_max_nc():
        h.nsplit = int(lib.pg_loss_reduce_parts(p.ptr(), p.ld, yp, yl, float(tconst), p.N, p.HW, p.C, buf.S.data_ptr(), st))
        if h.nsplit < 1:
            L.check(h.nsplit, 'pg_loss_reduce_parts')
        return h
    h.nsplit = 0          # staged: S is the combined result
    L.check(lib.pg_loss_reduce(p.ptr(), p.ld, yp, yl, float(tconst), p.N, p.HW, p.C, buf.S.data_ptr(), st), 'pg_loss_reduce')
    if need_sums:      # the two batch-global terms: only focal-Tversky and weighted BCE read them
        L.check(lib.pg_loss_prepare(buf.S.data_ptr(), p.N, p.C, beta, buf.sums.data_ptr(), st), 'pg_loss_prepare')
        h.sums = buf.sums
    if exchange:
        h.wait = allreduce(buf.sums)
    return h


def loss_finish(h, mode, alpha, grad_out, loss_out, loss_slot, bglobal, gamma=0.75):
    """Phase 2: the loss value into loss_out[loss_slot] and, if grad_out is not None, its gradient wrt p into View
    grad_out (seeded with the GLOBAL batch terms).  Returns the scratch buffers (see LossBuffers: not a result)."""
    lib = L.load()
    p, y, tconst, beta = h.args
    buf = h.buf
    if h.wait is not None:
        h.wait()
    st = _stream()
    yp, yl = (y.ptr(), y.ld) if y is not None else (None, 0)
    sums = h.sums.data_ptr() if h.sums is not None else None
    if p.N * p.C <= _fused_loss_max_nc():
        gp, gl = (grad_out.ptr(), grad_out.ld) if grad_out is not None else (None, 0)
        L.check(lib.pg_loss_value_grad(buf.S.data_ptr(), max(h.nsplit, 1), None, sums, mode, p.N, p.C, p.HW, bglobal, alpha, beta, gamma,
                                       p.ptr(), p.ld, yp, yl, float(tconst), gp, gl, L.ptr(loss_out, loss_slot), st), 'pg_loss_value_grad')
        return buf
    L.check(lib.pg_loss_finalize(buf.S.data_ptr(), sums, mode, p.N, p.C, p.HW, bglobal, alpha, beta, gamma,
                                 buf.coef.data_ptr(), L.ptr(loss_out, loss_slot), st), 'pg_loss_finalize')
    if grad_out is not None:
        gmode = {L.LOSS_TVERSKY: 0, L.LOSS_WBCE: 1, L.LOSS_BCE: 1, L.LOSS_MAE: 2}[mode]
        L.check(lib.pg_loss_grad(p.ptr(), p.ld, yp, yl, float(tconst), buf.coef.data_ptr(), grad_out.ptr(), grad_out.ld, p.N, p.HW, p.C,
                                 gmode, st), 'pg_loss_grad')
    return buf


def loss_value_and_grad(p, y, tconst, mode, alpha, grad_out, loss_out, loss_slot, bglobal, beta=0.75, gamma=0.75,
                        allreduce=None):
    """Evaluate one loss term over prediction View p against target View y (or the constant tconst), write its
    value into loss_out[loss_slot] and, if grad_out is not None, its gradient wrt p into View grad_out.
    `allreduce(tensor)` starts the SUM of the two global reduction terms across ranks and returns wait()."""
    h = loss_begin(p, y, tconst, beta, allreduce, need_sums=mode in (L.LOSS_TVERSKY, L.LOSS_WBCE))
    return loss_finish(h, mode, alpha, grad_out, loss_out, loss_slot, bglobal, gamma)
