"""Phase timing of the bf16 weight-gradient kernel k_wgrad_bf16x (diagnostics build: `make -C patchgan_amd/csrc trace`): per wave and
64-pixel chunk the time issuing the DMA pieces, waiting for them (+ barrier), the transposed reads + MFMAs, the closing barrier.
    PATCHGAN_LIB=patchgan_amd/libpatchgan_hip_trace.so python tools/trace_wg.py N Hb Wb Ca Cb stride"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchgan_amd import engine as E, _lib as L

N, Hb, Wb, Ca, Cb, s = (int(a) for a in sys.argv[1:7])
dev = torch.device('cuda')
op = E.ConvOp(N, Hb, Wb, Ca, Cb, s, L.ALGO_BF16)
big = E.View.alloc(N, Hb, Wb, Cb, dev, bf=True)
small = E.View.alloc(N, op.Hs, op.Ws, Ca, dev, bf=True)
big.t.normal_()
small.t.normal_()
dP = torch.zeros(16 * Ca * Cb, device=dev)
print(op.describe(2, L.IO_MASK))
lib = L.load()
lib.pg_debug_trace_set.argtypes = [ctypes.c_void_p]
buf = torch.zeros(16384 * 4 * 8, dtype=torch.int64, device=dev)
for _ in range(3):
    op.wgrad(small, big, dP, 0)
torch.cuda.synchronize()
assert lib.pg_debug_trace_set(ctypes.c_void_p(buf.data_ptr())) == 0
junk = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
junk.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
op.wgrad(small, big, dP, 0)
e1.record()
torch.cuda.synchronize()
r = buf.view(-1, 8).cpu()
r = r[r[:, 5] != 0].double()
tick = 0.01
span = float(r[:, 5].max() - r[:, 0].min())
print(f'{r.shape[0]} waves, whole call {1e3 * e0.elapsed_time(e1):.1f} us by events (slab reduce included), kernel {span * tick:.1f} us first entry to last exit')
nch = r[:, 7].clamp_min(1)
med = lambda x: float(x.median()) * tick
print(f'chunks per workgroup {int(nch.median())}')
print(f'per chunk: issue DMA pieces      {med(r[:, 1] / nch):8.3f} us')
print(f'per chunk: wait + barrier        {med(r[:, 2] / nch):8.3f} us')
print(f'per chunk: reads + MFMAs         {med(r[:, 3] / nch):8.3f} us')
print(f'per chunk: closing barrier       {med(r[:, 6] / nch):8.3f} us')
print(f'main loop in all                 {med(r[:, 4] - r[:, 0]):8.2f} us')
print(f'epilogue                         {med(r[:, 5] - r[:, 4]):8.2f} us')
