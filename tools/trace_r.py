"""Phase timing of k_conv_bf16r inside one launch (diagnostics build: `make -C patchgan_amd/csrc trace`): per wave the time (s_memrealtime, 10 ns)
from entry to the first barrier passed (first window + fragments), the waits at the later chunk starts, the time issuing a chunk's
MFMAs, the epilogue -- medians over all waves, in microseconds.
    PATCHGAN_LIB=patchgan_amd/libpatchgan_hip_trace.so python tools/trace_r.py N Hb Wb Ca Cb dir [mul]"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchgan_amd import engine as E, _lib as L

N, Hb, Wb, Ca, Cb, d = (int(a) for a in sys.argv[1:7])
mul = len(sys.argv) > 7
dev = torch.device('cuda')
op = E.ConvOp(N, Hb, Wb, Ca, Cb, 2, L.ALGO_BF16)
big = E.View.alloc(N, Hb, Wb, Cb, dev, bf=True)
small = E.View.alloc(N, op.Hs, op.Ws, Ca, dev, bf=True)
big.t.normal_()
small.t.normal_()
tv = E.View.alloc(N, Hb, Wb, Cb, dev, bf=True)
tv.t.uniform_(-1, 1)
P = torch.randn(16 * Ca * Cb, device=dev) * 0.02
print(op.describe(d, L.IO_MASK))
lib = L.load()
lib.pg_debug_trace_set.argtypes = [ctypes.c_void_p]
nwg = 1 << 16
buf = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=dev)


def run():
    if d == 0:
        op.big2small(big, P, 0, None, 0, small)
    elif mul:
        op.small2big(small, P, 0, None, 0, big, mul=(tv, L.ACT_TANH))
    else:
        op.small2big(small, P, 0, None, 0, big)


for _ in range(3):
    run()
torch.cuda.synchronize()
assert lib.pg_debug_trace_set(ctypes.c_void_p(buf.data_ptr())) == 0
junk = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
junk.zero_()                      # cold operands, as inside the step
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1)
r = buf.view(-1, 8).cpu()
r = r[r[:, 5] != 0].double()
tick = 0.01                       # s_memrealtime: 100 MHz
span = float(r[:, 5].max() - r[:, 0].min())
print(f'{r.shape[0]} waves, launch {us:.1f} us by events, {span * tick:.1f} us between the first entry and the last exit')
nch = r[:, 7]
med = lambda x: float(x.median()) * tick
print(f'chunks per tile {int(nch.median())}')
print(f'entry -> first barrier passed   {med(r[:, 1] - r[:, 0]):8.2f} us')
print(f'wait at a later chunk start     {med(r[:, 2] / (nch - 1).clamp_min(1)):8.2f} us  (x {int(nch.median()) - 1})')
print(f'issuing one chunk (64 MFMAs)    {med(r[:, 3] / nch):8.2f} us  (x {int(nch.median())})')
print(f'main loop in all                {med(r[:, 4] - r[:, 1]):8.2f} us')
print(f'epilogue                        {med(r[:, 5] - r[:, 4]):8.2f} us')
print(f'tile life                       {med(r[:, 5] - r[:, 0]):8.2f} us')
# workgroups in flight over time: sweep over entry / exit events of wave 0 of every workgroup
w0 = r[0::4] if r.shape[0] % 4 == 0 else r
ev = sorted([(float(t), 1) for t in w0[:, 0]] + [(float(t), -1) for t in w0[:, 5]])
cur, area, last = 0, 0.0, ev[0][0]
for tt, dlt in ev:
    area += cur * (tt - last)
    last = tt
    cur += dlt
print(f'workgroups in flight (time average) {area / span:.0f} of {w0.shape[0]}')
