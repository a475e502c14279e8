"""Parity at the shapes of BASELINE.json's remaining configurations, on the GPU through the C ABI:
  * cfg4: 512x512x3 -> 4-class masks (Softmax head, reference unet.py:48-49; weighted BCE over C > 1, trainer.py:75-80),
    nf = ndf = 64, fp32 and bf16 -- two training steps against the CPU oracle and its float64 run;
  * cfg5: 1024x1024 image -> 25 tiles of 256x256 (overlap 0.9, infer.py:155-174) through predict_image with nf = 64, against
    the oracle's n_crop / unet_forward / build_mask; tiles/s and peak VRAM are printed;
  * cfg2 at full width: step-1 weight gradients of the big layers against the oracle's autograd gradients.
Needs an MI355X."""
import time

import numpy as np
import pytest
import torch

from oracle import patchgan_oracle as O
from tests.golden_util import LOSS_KEYS

pytestmark = pytest.mark.gpu


def _rel(got, want):
    got, want = torch.as_tensor(got).double().cpu(), torch.as_tensor(want).double().cpu()
    return ((got - want).abs().max() / want.abs().max().clamp_min(1e-30)).item()


def _models(out_nc, final_act, seed=1234, nf=64, ndf=64):
    import patchgan_amd as pg
    torch.manual_seed(seed)
    g = pg.UNet(3, out_nc, nf, use_dropout=False, activation='leakyrelu', final_act=final_act)
    d = pg.Discriminator(3 + out_nc, ndf, n_layers=3)
    gw = {k: v.clone() for k, v in g.state_dict().items()}
    dw = {k: v.clone() for k, v in d.state_dict().items()}
    return g, d, gw, dw


def _inputs(B, out_nc, size, seed=7):
    gen = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, size, size, generator=gen)
    y = (torch.rand(B, out_nc, size, size, generator=gen) > 0.7).float()
    return x, y


def _curve(trainer, x, y, steps):
    rows = []
    for _ in range(steps):
        l = trainer.batch(x, y, train=True)
        rows.append([float(l[k]) for k in LOSS_KEYS])
    return np.array(rows)


def _relrows(a, b):
    return (np.abs(a - b) / np.maximum(np.abs(b), 1e-3)).max(axis=1)


@pytest.fixture(scope='module')
def cfg4_reference():
    """Oracle trajectories of the cfg4-shaped problem (B = 2 keeps the CPU oracle to ~15 s per step): fp32 on the CPU (the
    reference's own arithmetic) and float64 on the GPU (torch double ops), two training steps each."""
    g, d, gw, dw = _models(4, 'softmax')
    x, y = _inputs(2, 4, 512)
    kw = dict(activation='leakyrelu', final_act='softmax', n_layers=3, norm=False, loss_type='weighted_bce')
    c64 = _curve(O.OracleTrainer({k: v.cuda() for k, v in gw.items()}, {k: v.cuda() for k, v in dw.items()},
                                 dtype=torch.float64, **kw), x.cuda(), y.cuda(), 2)
    c32 = _curve(O.OracleTrainer(gw, dw, **kw), x, y, 2)
    with torch.no_grad():
        out0 = O.unet_forward(gw, x, 'leakyrelu', 'softmax')
    return dict(gw=gw, dw=dw, x=x, y=y, c32=c32, c64=c64, out0=out0)


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_cfg4_shape_two_steps_vs_oracle(cfg4_reference, precision, tmp_path):
    """Stated tolerances.  fp32: generator output at the initial weights within 2e-4 (max-norm) of the CPU oracle; each of
    the six loss scalars of both steps within 1e-4 of the float64 trajectory, and within max(1e-4, 4 x E) of the fp32 CPU
    oracle, E = that oracle's own distance from float64 (at 512x512 oneDNN's fp32 trajectory leaves float64 by ~3e-4 at
    step 2, more than the HIP path does).  bf16 (bf16 multiplies, fp32 accumulation / statistics / master weights): output
    within 2e-2, losses within 5e-2 of float64."""
    import patchgan_amd as pg
    from patchgan_amd import _lib as L
    r = cfg4_reference
    g, d, _, _ = _models(4, 'softmax')
    g.load_state_dict(r['gw'])
    d.load_state_dict(r['dw'])
    g.cuda()
    d.cuda()
    if precision == 'bf16':
        g.set_precision('bf16')
        d.set_precision('bf16')
    # the planner really runs the Winograd / fast kernels at these shapes (bs 2, 512x512)
    enc_ops, dec_ops = g.engine.ops(2, 512, 512)
    names = [op.describe(0)[0] for op in enc_ops] + [op.describe(1)[0] for op in dec_ops]
    if precision == 'fp32':
        assert sum(n.startswith('k_wino_bgemm') for n in names) >= 3, names
        assert any(op.describe(0)[0].startswith('k_wino_gemm') for op in d.engine.ops(4, 512, 512)), 'stride-1 Winograd'
    else:
        assert sum('bf16' in n for n in names) >= 8, names
    with torch.no_grad():
        g.train()
        out0 = g(r['x'].cuda()).cpu()
    e_out = _rel(out0, r['out0'])
    t = pg.Trainer(g, d, str(tmp_path / 'c'))
    t.loss_type = 'weighted_bce'
    t.setup_optimizers(1e-3, 1e-3)
    d.train()
    got = _curve(t, r['x'], r['y'], 2)
    e64, e32, env = _relrows(got, r['c64']), _relrows(got, r['c32']), _relrows(r['c32'], r['c64'])
    print(f'cfg4-shaped {precision}: output vs CPU oracle {e_out:.2e}; losses vs float64 {e64}, vs fp32 CPU {e32}; '
          f'fp32 CPU oracle vs float64 {env}')
    if precision == 'fp32':
        assert e_out < 2e-4
        assert (e64 <= 1e-4).all(), e64
        assert (e32 <= np.maximum(1e-4, 4 * env)).all(), (e32, env)
    else:
        assert 1e-6 < e_out < 2e-2
        assert (e64 <= 5e-2).all(), e64


def test_cfg5_tiled_inference_1024_vs_oracle():
    """BASELINE config 5 at full size: a 1024x1024 image is cut into 25 tiles of 256x256 (overlap 0.9), pushed through the
    nf = 64 generator and blended.  Against the oracle: tile extraction bit-exact, per-tile forward within 2e-4, the
    overlap-averaged probability map within 2e-4, and the thresholded mask equal except where |p - 0.5| < 1e-5."""
    from patchgan_amd.infer import predict_image
    from patchgan_amd import engine as E
    g, _, gw, _ = _models(1, 'sigmoid')
    g.cuda().eval()
    img = torch.rand(3, 1024, 1024, generator=torch.Generator().manual_seed(5))
    dimg = img.cuda()
    tiles = E.tiles_gather(dimg, 256, 0.9)
    assert tiles.N == 25
    crops = O.n_crop(img, 256, 0.9)
    assert torch.equal(tiles.to_nchw().cpu(), crops)
    with torch.no_grad():
        want_tiles = O.unet_forward(gw, crops, 'leakyrelu', 'sigmoid')
        got_tiles = g(crops.cuda()).cpu()
    assert _rel(got_tiles, want_tiles) < 2e-4
    prob = predict_image(g, dimg, 256, 0.9, 0.0)
    want_prob = O.build_mask(want_tiles.numpy(), 256, (1024, 1024), 0.0, 0.9)
    assert prob.shape == (1024, 1024) and prob.dtype == want_prob.dtype
    assert np.abs(prob - want_prob).max() < 2e-4
    mask = predict_image(g, dimg, 256, 0.9, 0.5)
    want_mask = O.build_mask(want_tiles.numpy(), 256, (1024, 1024), 0.5, 0.9)
    differ = mask != want_mask
    assert differ.mean() < 1e-3 and (np.abs(want_prob[differ] - 0.5) < 1e-5).all()
    # throughput + memory of the streaming path (informational; D2H copy of the mask included)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(10):
        predict_image(g, dimg, 256, 0.9, 0.5)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f'cfg5: 1024x1024 -> 25 tiles, nf=64 fp32: {dt * 1e3:.2f} ms per image = {25 / dt:.0f} tiles/s, '
          f'peak VRAM {torch.cuda.max_memory_allocated() / 2 ** 30:.2f} GiB')


def test_cfg2_full_width_gradients_vs_oracle(tmp_path):
    """Step-1 weight gradients at the benchmark layer shapes (nf = ndf = 64, 256x256; B = 4 keeps the CPU oracle to seconds):
    EVERY generator and discriminator parameter gradient against the oracle run in float64 (torch double ops on the GPU),
    next to the fp32 CPU oracle's own distance from float64.

    Stated tolerances.  Discriminator (well conditioned; its d3 / d2 / d1 layers run the Winograd kernels): relative max-norm
    1e-4 (measured <= 1.2e-5; the fp32 CPU oracle: 1e-5 .. 7e-5).  Generator: relative L2 2e-2 and no worse than 8 x the fp32
    CPU oracle's own relative-L2 distance + 1e-3 -- at this width the generator's backward chain amplifies fp32 rounding so
    much that EXACT fp32 evaluations (the one-thread-per-output kernels, the implicit GEMM, oneDNN on the CPU) sit 3e-4 .. 7e-3
    in relative L2 and up to 1e-1 in max-norm from float64 (tools/debug_grads_full.py prints the table), so a max-norm bound
    would test the conditioning of the network, not the kernels; the per-kernel tests carry the 2e-5 statements."""
    import patchgan_amd as pg
    g, d, gw, dw = _models(1, 'sigmoid')
    x, y = _inputs(4, 1, 256)
    kw = dict(activation='leakyrelu', final_act='sigmoid', n_layers=3, norm=False, loss_type='tversky')
    ot = O.OracleTrainer(gw, dw, **kw)
    want_l = ot.batch(x, y, train=True)
    o64 = O.OracleTrainer({k: v.cuda() for k, v in gw.items()}, {k: v.cuda() for k, v in dw.items()}, dtype=torch.float64, **kw)
    o64.batch(x.cuda(), y.cuda(), train=True)
    t = pg.Trainer(g.cuda(), d.cuda(), str(tmp_path / 'c'))
    t.setup_optimizers(1e-3, 1e-3)
    g.train()
    d.train()
    got_l = t.batch(x, y, train=True)
    for k in LOSS_KEYS:
        assert abs(got_l[k] - want_l[k]) <= 1e-4 * max(abs(want_l[k]), 1e-3), (k, got_l[k], want_l[k])

    def l2(a, b):
        a, b = a.double().cpu(), b.double().cpu()
        return ((a - b).norm() / b.norm()).item()

    rows = []
    for k, p in d.named_parameters():
        e = _rel(p.grad, o64.last['d_grads'][k])
        rows.append(('D', k, e, _rel(ot.last['d_grads'][k], o64.last['d_grads'][k])))
        assert e < 1e-4, (k, e)
    for k, p in g.named_parameters():
        e, noise = l2(p.grad, o64.last['g_grads'][k]), l2(ot.last['g_grads'][k], o64.last['g_grads'][k])
        rows.append(('G', k, e, noise))
        assert e < 2e-2 and e <= 8 * noise + 1e-3, (k, e, noise)
    print('cfg2 full-width gradients vs float64 (D: relative max-norm, G: relative L2) | fp32 CPU oracle vs float64:')
    for net, k, e, n in rows:
        print(f'   {net} {k:38s} {e:.1e} | {n:.1e}')
    keys = [k for _, k, _, _ in rows]
    for key in ('encoder.3.model.DownConv3.weight', 'decoder.3.model.UpConv3.weight', 'model.6.weight'):
        assert key in keys       # the big layers named in the review are among those compared
