"""Where does the distance between fp32 evaluations of the cfg2 loss curve come from?  ("weak" item 1.)

At the benchmark configuration (cfg2: bs 16, 256 x 256, nf = ndf = 64) every fp32 evaluation of the 10-step loss curve leaves
the float64 trajectory by ~1e-4 around steps 4-6: oneDNN (the reference) 0.84e-4, this library without Winograd 1.25e-4, with
Winograd 1.43e-4.  This script separates the PROBLEM's conditioning from any one kernel's rounding, on one GPU:

  A. conditioning: the float64 oracle re-run from initial weights perturbed by one fp32 ulp (w * (1 +- 2^-24), random signs; three
     seeds) -- how far does the float64 trajectory itself move?  That is what rounding the WEIGHTS to fp32 once is worth, before any
     fp32 arithmetic at all.
  B. one stage in fp32 at a time inside the float64 oracle (operands rounded to fp32, the op in torch's fp32 GPU kernels, result
     widened again; autograd carries the same casts into the backward pass): convolutions | InstanceNorm | activations | losses |
     Adam + fp32 weight storage.  The distance each run ends up from float64 is that stage's share.
  C. the complete fp32 evaluations next to each other: the reference's own curve (tests/golden/w_cfg2.npz), the fp32 oracle on
     torch-GPU kernels (MIOpen), the HIP path (default), the HIP path with every Winograd kernel off, with only the stride-1
     Winograd off, with only the weight-gradient Winograd off.

Round 6: --config cfg2 | cfg4 | cfg1 picks the fixture (w_cfg2: the benchmark configuration; w_cfg4: cfg4's shape, 512 x 512, 4 classes, softmax +
weighted BCE, B = 2, 4 steps; w_cfg1: the COCO yaml's hyper-parameters, relu + weighted BCE + a 5-layer discriminator, B = 4, 10 steps); the HIP
rows include the split-bf16 GEMMs off (PG_TUNE_S3_OFF); --hip-only prints just the default HIP row against float64 and the reference (for
process-wide experiment switches such as PATCHGAN_S3_VAR, which are read once per process).

Usage on the GPU box:  python tools/parity_attribution.py [--config cfgN] [--hip-only] [steps] > gpurun_out/parity_attribution.txt"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

import patchgan_amd as pg
from patchgan_amd import _lib as L
from oracle import patchgan_oracle as O
from tests.golden_util import Golden, LOSS_KEYS

argv = sys.argv[1:]
config = 'cfg2'
if '--config' in argv:
    i = argv.index('--config')
    config = argv[i + 1]
    del argv[i:i + 2]
hip_only = '--hip-only' in argv
if hip_only:
    argv.remove('--hip-only')
gold = Golden('w_' + config)
c = gold.cfg
steps = min(int(argv[0]) if argv else 10, gold.nsteps)
print(f'fixture w_{config}: B = {c["B"]}, {c["size"]} x {c["size"]}, nf = {c["nf"]}, ndf = {c["ndf"]}, n_layers = {c["n_layers"]}, '
      f'{c["activation"]} / {c["final_act"]}, {c["loss_type"]}, {steps} steps')
g0, d0 = gold.seeded_modules()
gw = {k: v.clone() for k, v in g0.state_dict().items()}
dw = {k: v.clone() for k, v in d0.state_dict().items()}
x, y = gold.inputs()
kw = dict(activation=c['activation'], final_act=c['final_act'], n_layers=c['n_layers'], norm=c['norm'], loss_type=c['loss_type'])
cuda = lambda w: {k: v.cuda() for k, v in w.items()}
np.set_printoptions(precision=2, linewidth=220)


def run(tr, xx, yy):
    rows = []
    for _ in range(steps):
        l = tr.batch(xx, yy, train=True)
        rows.append([float(l[k]) for k in LOSS_KEYS])
    return np.array(rows)


def rel(a, b):
    return (np.abs(a - b) / np.maximum(np.abs(b), 1e-3)).max(axis=1)


def report(name, curve, base):
    e = rel(curve, base)
    print(f'{name:58s} max {e.max():.2e} | per step {e}', flush=True)
    return e


c64 = run(O.OracleTrainer(cuda(gw), cuda(dw), dtype=torch.float64, **kw), x.cuda(), y.cuda())
print('float64 oracle (torch double on the GPU): the yardstick.  gdisc per step', c64[:, 2], flush=True)

if not hip_only:
    print('\nA. conditioning -- float64 arithmetic, initial weights perturbed by one fp32 ulp')
    for seed in (1, 2, 3):
        gen = torch.Generator().manual_seed(seed)

        def jiggle(w):
            out = {}
            for k, v in w.items():
                s = (torch.randint(0, 2, v.shape, generator=gen).double() * 2 - 1) * 2.0 ** -24
                out[k] = (v.double() * (1 + s)).cuda()
            return out
        tr = O.OracleTrainer(jiggle(gw), jiggle(dw), dtype=torch.float64, **kw)
        report(f'  float64, weights * (1 +- 2^-24), seed {seed}', run(tr, x.cuda(), y.cuda()), c64)

    print('\nB. ONE stage in fp32 inside the float64 oracle')


    class _FProxy:
        """torch.nn.functional with some functions evaluated in fp32 (operands rounded, result widened)."""

        def __init__(self, names):
            self._names = set(names)

        def __getattr__(self, n):
            fn = getattr(F, n)
            if n not in self._names:
                return fn

            def lowp(*a, **k):
                a = [t.float() if torch.is_tensor(t) and t.dtype == torch.float64 else t for t in a]
                k = {kk: (t.float() if torch.is_tensor(t) and t.dtype == torch.float64 else t) for kk, t in k.items()}
                return fn(*a, **k).double()
            return lowp


    def with_stage(stage):
        saved = (O.F, O.apply_act, O.seg_loss, O.bce, O.adam_update)
        try:
            if stage == 'conv':
                O.F = _FProxy(['conv2d', 'conv_transpose2d'])
            elif stage == 'instnorm':
                O.F = _FProxy(['instance_norm'])
            elif stage == 'act':
                act = O.apply_act
                O.apply_act = lambda t, name: act(t.float(), name).double()
            elif stage == 'loss':
                seg, b = O.seg_loss, O.bce
                O.seg_loss = lambda lt, p, t, *a, **k: seg(lt, p.float(), t.float(), *a, **k).double()
                O.bce = lambda p, t, weight=None: b(p.float(), t.float(), weight=weight.float() if weight is not None else None).double()
            elif stage == 'adam':
                upd = O.adam_update

                def adam32(p, g, m, v, t, lr, *a, **k):
                    p32, g32, m32, v32 = p.float(), g.float(), m.float(), v.float()
                    upd(p32, g32, m32, v32, t, lr, *a, **k)
                    p.copy_(p32), m.copy_(m32), v.copy_(v32)
                O.adam_update = adam32
            tr = O.OracleTrainer(cuda(gw), cuda(dw), dtype=torch.float64, **kw)
            return run(tr, x.cuda(), y.cuda())
        finally:
            O.F, O.apply_act, O.seg_loss, O.bce, O.adam_update = saved


    for stage, label in (('conv', 'convolutions fwd + bwd in fp32 (MIOpen)'), ('instnorm', 'InstanceNorm fwd + bwd in fp32'),
                         ('act', 'activations fwd + bwd in fp32'), ('loss', 'the four loss terms in fp32'),
                         ('adam', 'Adam + weight storage in fp32')):
        report('  float64 except ' + label, with_stage(stage), c64)

print('\nC. complete fp32 evaluations vs float64')
ref = gold.z['losses'][:steps]
e_ref = report(f'  the REFERENCE itself (tests/golden/w_{config}.npz, oneDNN 8 thr)', ref, c64)
if not hip_only:
    report('  fp32 oracle on torch-GPU kernels (MIOpen)', run(O.OracleTrainer(cuda(gw), cuda(dw), **kw), x.cuda(), y.cuda()), c64)
hip = {}
variants = (('HIP default', 0),) if hip_only else (
    ('HIP default', 0), ('HIP, split-bf16 GEMMs off (fp32 MFMA everywhere)', L.TUNE_S3_OFF), ('HIP, every Winograd kernel off', L.TUNE_WINO_OFF),
    ('HIP, weight-gradient Winograd off', L.TUNE_WINOW_OFF | L.TUNE_WINO2W_OFF),
    ('HIP, polyphase (stride-2) Winograd off', L.TUNE_WINO2_OFF | L.TUNE_WINO2W_OFF))
for label, bits in variants:
    g = pg.UNet(c['in_nc'], c['out_nc'], c['nf'], use_dropout=False, activation=c['activation'], final_act=c['final_act'])
    d = pg.Discriminator(c['in_nc'] + c['out_nc'], c['ndf'], n_layers=c['n_layers'])
    g.load_state_dict(gw)
    d.load_state_dict(dw)
    g.set_tuning(bits)
    d.set_tuning(bits)
    t = pg.Trainer(g.cuda(), d.cuda(), tempfile.mkdtemp())
    t.loss_type = c['loss_type']
    t.setup_optimizers(1e-3, 1e-3)
    g.train()
    d.train()
    hip[label] = run(t, x, y)
    report('  ' + label, hip[label], c64)
    del t, g, d
print('\nD. the HIP path vs the REFERENCE curve (what the parity test asserts)')
for label, curve in hip.items():
    report('  ' + label + ' vs reference', curve, ref)
print('\nper-step table, HIP default | reference | float64:')
for s in range(steps):
    print(f'  step {s + 1:2d}', ' '.join(f'{k}: {hip["HIP default"][s, i]:.6f} | {ref[s, i]:.6f} | {c64[s, i]:.6f}' for i, k in enumerate(LOSS_KEYS) if k != 'gen_loss'))
