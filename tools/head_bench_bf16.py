"""small -> big of the few-channel heads from a bf16 `small` (fp32 result): microseconds per call.  PATCHGAN_EXPERIMENT=1 PATCHGAN_NO_TAPNF=1
restores the bf16 row GEMM + col2im.  usage: python tools/head_bench_bf16.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchgan_amd import engine as E, _lib as L

dev = torch.device('cuda')
for name, geom in (('cfg2 dec6 fwd', (16, 256, 256, 128, 1, 2)), ('cfg2 d0/N dgrad', (16, 256, 256, 64, 4, 2)),
                   ('cfg4 dec6 fwd', (8, 512, 512, 128, 4, 2)), ('cfg4 d0/N dgrad', (8, 512, 512, 64, 7, 2))):
    N, Hb, Wb, Ca, Cb, s = geom
    op = E.ConvOp(*geom, L.ALGO_BF16)
    small = E.View.alloc(N, op.Hs, op.Ws, Ca, dev, bf=True)
    small.t.normal_()
    big = E.View.alloc(N, Hb, Wb, Cb, dev)
    P = torch.randn(16 * Ca * Cb, device=dev) * 0.05
    fn = lambda: op.small2big(small, P, 0, None, 0, big)
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f'{name:18s} {str(geom):32s} {e0.elapsed_time(e1) * 50:7.1f} us  {op.describe(1, L.IO_SMALL_BF16)[0]}')
