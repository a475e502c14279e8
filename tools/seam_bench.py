"""Timing of the image-facing (few-channel) conv layers: tools/seam_bench.py  (PATCHGAN_EXPERIMENT=1 PATCHGAN_NO_TAPK=1 for the generic kernels)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchgan_amd import engine as E, _lib as L
dev = torch.device('cuda')
for algo, tag in ((L.ALGO_AUTO, 'auto'), (L.ALGO_BF16, 'bf16')):
    for (N, H, Ca, Cb) in ((8, 512, 64, 3), (8, 512, 64, 7), (16, 512, 64, 7), (8, 512, 64, 8), (8, 512, 64, 5), (16, 256, 64, 4), (32, 256, 64, 4)):
        op = E.ConvOp(N, H, H, Ca, Cb, 2, algo)
        big = E.View.alloc(N, H, H, Cb, dev); big.t.normal_()
        small = E.View.alloc(N, op.Hs, op.Ws, Ca, dev)
        P = torch.randn(16 * Ca * Cb, device=dev) * 0.05
        dP = torch.empty_like(P)
        small.t.normal_()
        for oc, fn in ((0, lambda: op.big2small(big, P, 0, None, 0, small, 1)), (2, lambda: op.wgrad(small, big, dP, 0))):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            mb = (big.t.numel() + small.t.numel()) * 4 / 1e6
            print(f"{tag} N={N} {H}x{H} {Cb}->{Ca} op{oc}: {ms*1e3:7.1f} us  {op.describe(oc)[0]:28s} {mb/ms/1e3:6.2f} TB/s of tensors", flush=True)
