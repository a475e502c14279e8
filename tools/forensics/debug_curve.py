"""Per-key loss-curve error of the HIP trainer against the float64 oracle for one golden config (fp32 oracle noise beside it).
usage: [PATCHGAN_ALGO=direct|mfma] python tools/debug_curve.py d_softmax_tversky"""
import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from tests.golden_util import Golden, LOSS_KEYS
from tests.test_step_gpu import oracle_curves, build

np.set_printoptions(linewidth=200, precision=2)
name = sys.argv[1] if len(sys.argv) > 1 else 'd_softmax_tversky'
gold = Golden(name)
import pathlib
g, d, t = build(gold, pathlib.Path(tempfile.mkdtemp()))
x, y = gold.inputs()
g.train(); d.train()
curve = []
for s in range(gold.nsteps):
    l = t.batch(x, y, train=True)
    curve.append([l[k] for k in LOSS_KEYS])
curve = np.array(curve)
c32, _ = oracle_curves(gold, torch.float32)
c64, _ = oracle_curves(gold, torch.float64)
rel = lambda a, b: np.abs(a - b) / np.maximum(np.abs(b), 1e-6)
print(LOSS_KEYS, 'algo', os.environ.get('PATCHGAN_ALGO', 'auto'))
print('hip vs f64\n', rel(curve, c64))
print('f32 oracle vs f64\n', rel(c32, c64))
