"""Debug aid (GPU box): per-layer deviation of the bf16 path from the fp32 path on a golden config."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, tempfile
import patchgan_amd as pg
from patchgan_amd import engine as E
from tests.golden_util import Golden
name = sys.argv[1] if len(sys.argv) > 1 else 'a_lrelu_tversky'
gold = Golden(name); c = gold.cfg
x, y = gold.inputs()
def build(prec):
    g = pg.UNet(c['in_nc'], c['out_nc'], c['nf'], activation=c['activation'], final_act=c['final_act'])
    g.load_state_dict(gold.weights('g0')); g.cuda().set_precision(prec); g.train()
    return g
outs = {}
for prec in ('fp32', 'bf16'):
    g = build(prec)
    N, C, H, W = x.shape
    xin = E.View.alloc(N, H, W, C, g.flat.device).from_nchw(x.cuda())
    gen = E.View.alloc(N, H, W, c['out_nc'], g.flat.device)
    ctx = g.engine.forward(g.flat, xin, gen, True, 0)
    outs[prec] = dict(y=[v.to_nchw() for v in ctx.y], enc=[v.to_nchw() for v in ctx.enc_out], gen=gen.to_nchw(),
                      algo=g.engine.algo, desc=[op.describe(0) for op in g.engine.ops(N, H, W)[0]])
rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
print('enc kernels', outs['bf16']['desc'])
for i in range(7):
    print(f"enc{i}: conv out rel {rel(outs['bf16']['y'][i], outs['fp32']['y'][i]):.3e}   block out rel {rel(outs['bf16']['enc'][i], outs['fp32']['enc'][i]):.3e}  shape {tuple(outs['fp32']['y'][i].shape)}")
print('gen rel', rel(outs['bf16']['gen'], outs['fp32']['gen']), 'abs max diff', (outs['bf16']['gen'] - outs['fp32']['gen']).abs().max().item())
