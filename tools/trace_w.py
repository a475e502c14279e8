"""Phase timing of the fp32 polyphase GEMM k_wino_bgemm inside one layer call (diagnostics build: `make -C patchgan_amd/csrc trace`):
per wave the prologue (first chunk staged), the time issuing a chunk's loads / LDS reads / MFMAs, the barriers + LDS staging between
chunks, the epilogue -- medians over all waves, microseconds (s_memrealtime, 10 ns).
    PATCHGAN_LIB=patchgan_amd/libpatchgan_hip_trace.so python tools/trace_w.py N Hb Wb Ca Cb dir"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchgan_amd import engine as E, _lib as L

N, Hb, Wb, Ca, Cb, d = (int(a) for a in sys.argv[1:7])
dev = torch.device('cuda')
op = E.ConvOp(N, Hb, Wb, Ca, Cb, 2, L.ALGO_AUTO)
big = E.View.alloc(N, Hb, Wb, Cb, dev)
small = E.View.alloc(N, op.Hs, op.Ws, Ca, dev)
big.t.normal_()
small.t.normal_()
P = torch.randn(16 * Ca * Cb, device=dev) * 0.02
print(op.describe(d, 0))
lib = L.load()
lib.pg_debug_trace_set_w.argtypes = [ctypes.c_void_p]
buf = torch.zeros(16384 * 4 * 8, dtype=torch.int64, device=dev)


def run():
    if d == 0:
        op.big2small(big, P, 0, None, 0, small)
    else:
        op.small2big(small, P, 0, None, 0, big)


for _ in range(3):
    run()
torch.cuda.synchronize()
assert lib.pg_debug_trace_set_w(ctypes.c_void_p(buf.data_ptr())) == 0
junk = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
junk.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
r = buf.view(-1, 8).cpu()
r = r[r[:, 5] != 0].double()
tick = 0.01
span = float(r[:, 5].max() - r[:, 0].min())
print(f'{r.shape[0]} waves, whole call {1e3 * e0.elapsed_time(e1):.1f} us by events (transforms included), GEMM {span * tick:.1f} us first entry to last exit')
nch = r[:, 7]
med = lambda x: float(x.median()) * tick
print(f'chunks per tile {int(nch.median())}')
print(f'prologue (first chunk staged)        {med(r[:, 1] - r[:, 0]):8.2f} us')
print(f'one chunk: loads + LDS reads + MFMAs {med(r[:, 3] / nch):8.2f} us  (x {int(nch.median())})')
print(f'one chunk: barriers + LDS staging    {med(r[:, 2] / nch):8.2f} us  (x {int(nch.median())})')
print(f'main loop in all                     {med(r[:, 4] - r[:, 1]):8.2f} us')
print(f'epilogue                             {med(r[:, 5] - r[:, 4]):8.2f} us')
print(f'tile life                            {med(r[:, 5] - r[:, 0]):8.2f} us')
w0 = r[0::4]
ev = sorted([(float(t), 1) for t in w0[:, 0]] + [(float(t), -1) for t in w0[:, 5]])
cur, area, last = 0, 0.0, ev[0][0]
for tt, dlt in ev:
    area += cur * (tt - last)
    last = tt
    cur += dlt
print(f'workgroups in flight (time average) {area / span:.0f} of {w0.shape[0]}')
