"""Step-1 parameter gradients and post-Adam parameters of the HIP trainer against the float64 oracle, per parameter, with the
fp32 oracle's own distance beside them (which parameter's gradient carries more rounding noise than torch's fp32 run?).
usage: python tools/debug_step1.py d_softmax_tversky"""
import sys, os, tempfile, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from oracle import patchgan_oracle as O
from tests.golden_util import Golden
from tests.test_step_gpu import build

name = sys.argv[1] if len(sys.argv) > 1 else 'd_softmax_tversky'
gold = Golden(name)
c = gold.cfg
x, y = gold.inputs()


def oracle(dtype):
    ot = O.OracleTrainer(gold.weights('g0'), gold.weights('d0'), activation=c['activation'], final_act=c['final_act'],
                         n_layers=c['n_layers'], norm=c['norm'], loss_type=c['loss_type'], dtype=dtype)
    ot.batch(x, y, train=True)
    return ot


o32, o64 = oracle(torch.float32), oracle(torch.float64)
g, d, t = build(gold, pathlib.Path(tempfile.mkdtemp()))
g.train(); d.train()
t.batch(x, y, train=True)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


for tag, net, grads32, grads64, w32, w64 in (('G', g, o32.last['g_grads'], o64.last['g_grads'], o32.gw, o64.gw),
                                             ('D', d, o32.last['d_grads'], o64.last['d_grads'], o32.dw, o64.dw)):
    print(f'{tag}: name | max|g| | grad err hip, f32 oracle (rel max-norm vs f64) | post-Adam param err hip, f32 (abs max)')
    for k, p in net.named_parameters():
        g64 = grads64[k]
        eh, e32 = rel(p.grad, g64), rel(grads32[k], g64)
        ph = float((p.detach().double().cpu() - w64[k].detach().double()).abs().max())
        p32 = float((w32[k].detach().double() - w64[k].detach().double()).abs().max())
        flag = ' <<<' if eh > 4 * max(e32, 1e-6) or ph > 4 * max(p32, 1e-6) else ''
        print(f'  {k:44s} {float(g64.abs().max()):9.2e} | {eh:8.1e} {e32:8.1e} | {ph:8.1e} {p32:8.1e}{flag}')
