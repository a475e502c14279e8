"""Stress of the discriminator head's data gradient with a bf16 output (k_s2b_ca1_s1 under PATCHGAN_CA1S1_BF16=1, else k_s2b_ca1):
ITERS launches on fixed operands, every output compared ON THE DEVICE with the first one; optional allocation churn in the same
process (CHURN=1: empty_cache + fresh allocations between launches, what a starting process does).  Run two at once.
usage: [PATCHGAN_EXPERIMENT=1 PATCHGAN_CA1S1_BF16=1] python tools/stress_ca1s1.py [iters] [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchgan_amd import engine as E, _lib as L

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1
churn = os.environ.get('CHURN') == '1'
dev = torch.device('cuda')
torch.manual_seed(0)
op = E.ConvOp(N, 31, 31, 1, 512, 1, L.ALGO_BF16)
small = E.View.alloc(N, 30, 30, 1, dev)
small.t.normal_()
t3 = E.View.alloc(N, 31, 31, 512, dev, bf=True)
t3.t.copy_(torch.tanh(torch.randn_like(t3.t.float())).bfloat16())
P = torch.randn(16 * 512, device=dev) * 0.05
out = E.View.alloc(N, 31, 31, 512, dev, bf=True)
print(op.describe(1, L.IO_BIG_BF16)[0], 'mul_ok', op.mul_ok(small, out, t3))
op.small2big(small, P, 0, None, 0, out, mul=(t3, L.ACT_CODES['tanh']))
ref = out.t.clone()
bad = torch.zeros((), dtype=torch.int64, device=dev)
x = torch.randn(1 << 22, device=dev)
for i in range(iters):
    out.t.zero_()
    x.mul_(1.0001)                                  # a neighbour kernel before ...
    op.small2big(small, P, 0, None, 0, out, mul=(t3, L.ACT_CODES['tanh']))
    x.add_(0.5)                                     # ... and after
    bad += (out.t.view(torch.int16) != ref.view(torch.int16)).any().to(torch.int64)
    if churn and i % 50 == 0:
        torch.cuda.empty_cache()
        junk = [torch.empty(1 << (20 + (i // 50) % 6), device=dev) for _ in range(4)]
        del junk
torch.cuda.synchronize()
print(f'pid {os.getpid()}: {int(bad)} of {iters} launches differ from the first')
