"""Per-kernel parity of the HIP path (through the C ABI) against the CPU oracle.  Needs an MI355X."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import patchgan_oracle as O

pytestmark = pytest.mark.gpu

# (N, Hb, Wb, Ca, Cb, stride)
GEOMS = [
    (2, 16, 16, 8, 4, 2),
    (2, 16, 16, 64, 32, 2),
    (1, 8, 8, 160, 96, 2),      # multi-tile N, K not tile multiple
    (3, 12, 20, 36, 20, 2),     # ragged M/N, non-square
    (2, 9, 9, 8, 8, 1),         # stride 1
    (2, 11, 7, 1, 24, 1),       # Ca = 1 (the D head)
    (2, 16, 16, 16, 3, 2),      # Cb = 3 (RGB input): scalar-K path
    (2, 16, 16, 12, 1, 2),      # Cb = 1 (mask output)
    (1, 15, 13, 8, 8, 2),       # odd big extent
    (2, 4, 4, 512, 64, 2),      # small M, long K: split-K
    (4, 64, 64, 32, 16, 2),     # many rows
    (2, 16, 16, 4, 24, 2),      # Ca <= 8, stride 2: wgrad taps-in-N mode 2
    (1, 15, 13, 3, 20, 2),      # same, odd extent, Ca % 4 != 0
    (2, 32, 32, 64, 4, 2),      # d0-like: wgrad taps-in-N mode 1, N = 64
    (2, 32, 32, 64, 32, 1),     # d3-like stride 1, Hs = 31 (carry-chain pixel decode in the fast wgrad)
    (3, 34, 38, 40, 36, 2),     # non-power-of-two extents >= 16, ragged channel tiles
    (2, 64, 64, 128, 64, 2),    # enc-like: all fast kernels, power-of-two decode
    (2, 16, 16, 1, 64, 1),      # D-head-like: big2small via row GEMM + tap gather
    (1, 15, 13, 4, 32, 2),      # same path, stride 2, odd extents, 4 output channels
    (2, 16, 16, 32, 1, 2),      # dec6-like: stride-2 small2big onto <= 4 channels from 32 / 64 / 128 in one pass (k_s2b_tapnf<1>)
    (2, 12, 20, 64, 3, 2),      # same kernel, 3 channels, blocks that overhang the image
    (3, 34, 30, 128, 4, 2),     # same, 4 channels (16-byte stores), 128 input channels, several blocks per sample
    (2, 16, 16, 96, 1, 2),      # other channel counts: small2big via row GEMM + col2im, Cb = 1 (no weight re-layout)
    (2, 12, 20, 96, 3, 2),      # same path with the (tap, b) weight re-layout, Cb = 3
    (2, 9, 9, 32, 8, 1),        # same path, stride 1
    (4, 64, 64, 64, 4, 2),      # >= 4096 small pixels on <= 4 big-side channels: the persistent image-facing kernels (k_b2s_tapkp, k_wgrad_tapnp)
    (5, 64, 64, 72, 3, 2),      # same, ragged channel blocks, 3-channel pixels
    (4, 66, 62, 132, 1, 2),     # same, one big-side channel (dec6's backward), ragged pixel tiles
    (4, 64, 64, 40, 2, 2),      # same, two channels
]
ACTS = {'none': 0, 'leakyrelu': 1, 'relu': 2, 'tanh': 3, 'sigmoid': 4}


def _mk(N, Hb, Wb, Ca, Cb, s, seed=0):
    g = torch.Generator().manual_seed(seed)
    Hs, Ws = (Hb - 2) // s + 1, (Wb - 2) // s + 1
    big = torch.randn(N, Cb, Hb, Wb, generator=g)
    small = torch.randn(N, Ca, Hs, Ws, generator=g)
    Wt = torch.randn(Ca, Cb, 4, 4, generator=g) / math.sqrt(Cb * 16)
    return big, small, Wt, Hs, Ws


@pytest.mark.parametrize('algo', [1, 2, 3], ids=['direct', 'mfma', 'bf16'])
@pytest.mark.parametrize('geom', GEOMS, ids=lambda g: 'x'.join(map(str, g)))
def test_big2small(geom, algo):
    from patchgan_amd import engine as E
    from tests.gpu_util import to_view, empty_view, pack, rel_err
    N, Hb, Wb, Ca, Cb, s = geom
    big, _, Wt, Hs, Ws = _mk(*geom)
    bias = torch.randn(Ca)
    op = E.ConvOp(N, Hb, Wb, Ca, Cb, s, algo)
    for act, ld_extra in (('none', 0), ('tanh', 4)):
        vb = to_view(big, ld=Cb + ld_extra, off=0)
        vs = empty_view(N, Hs, Ws, Ca, ld=Ca + ld_extra, off=ld_extra // 2)
        P = pack(Wt)
        op.big2small(vb, P, 0, bias.cuda(), 0, vs, ACTS[act])
        want = O.apply_act(F.conv2d(big, Wt, bias, stride=s, padding=1), act)
        got = vs.to_nchw()
        torch.cuda.synchronize()
        assert rel_err(got, want) < (2e-2 if algo == 3 else 2e-5), (geom, act)


@pytest.mark.parametrize('algo', [1, 2, 3], ids=['direct', 'mfma', 'bf16'])
@pytest.mark.parametrize('geom', GEOMS, ids=lambda g: 'x'.join(map(str, g)))
def test_small2big(geom, algo):
    from patchgan_amd import engine as E
    from tests.gpu_util import to_view, empty_view, pack, rel_err
    N, Hb, Wb, Ca, Cb, s = geom
    _, small, Wt, Hs, Ws = _mk(*geom)
    bias = torch.randn(Cb)
    op = E.ConvOp(N, Hb, Wb, Ca, Cb, s, algo)
    for act, ld_extra in (('none', 0), ('sigmoid', 4)):
        vs = to_view(small, ld=Ca + ld_extra)
        vb = empty_view(N, Hb, Wb, Cb, ld=Cb + ld_extra, off=ld_extra // 4)
        op.small2big(vs, pack(Wt), 0, bias.cuda(), 0, vb, ACTS[act])
        opad = (Hb - ((Hs - 1) * s + 2), Wb - ((Ws - 1) * s + 2))
        want = F.conv_transpose2d(small, Wt, bias, stride=s, padding=1, output_padding=opad)
        assert want.shape[2:] == (Hb, Wb)
        want = O.apply_act(want, act)
        got = vb.to_nchw()
        torch.cuda.synchronize()
        assert rel_err(got, want) < (2e-2 if algo == 3 else 2e-5), (geom, act)


@pytest.mark.parametrize('algo', [1, 2, 3], ids=['direct', 'mfma', 'bf16'])
@pytest.mark.parametrize('geom', GEOMS, ids=lambda g: 'x'.join(map(str, g)))
def test_wgrad(geom, algo):
    from patchgan_amd import engine as E
    from tests.gpu_util import to_view, unpack, rel_err, DEV
    N, Hb, Wb, Ca, Cb, s = geom
    big, small, Wt, Hs, Ws = _mk(*geom)
    op = E.ConvOp(N, Hb, Wb, Ca, Cb, s, algo)
    Wr = Wt.clone().requires_grad_(True)
    br = torch.zeros(Ca, requires_grad=True)
    F.conv2d(big, Wr, br, stride=s, padding=1).backward(small)
    dP = torch.full((16 * Ca * Cb,), float('nan'), device=DEV)
    db = torch.full((Ca + 4,), float('nan'), device=DEV)
    op.wgrad(to_view(small, ld=Ca + 4, off=4), to_view(big, ld=Cb + 8, off=4), dP, 0, db, 0)
    torch.cuda.synchronize()
    assert rel_err(unpack(dP, Ca, Cb), Wr.grad) < (2e-2 if algo == 3 else 3e-5), geom
    assert rel_err(db[:Ca], br.grad) < 3e-5, geom


# stride-1 layers wide enough for the Winograd F(2x2, 4x4) path that PG_ALGO_AUTO selects (>= 2048 tiles, channels >= 64,
# Cin % 32 == 0): odd / even extents (ragged last tile row and column), ragged channel tile, both directions
WINO_GEOMS = [(9, 32, 32, 128, 64, 1), (3, 55, 58, 96, 64, 1), (10, 31, 31, 64, 160, 1)]


@pytest.mark.parametrize('tile', ['auto', 'f2', 'f3', 'f3_dma', 'f2_dma'])
@pytest.mark.parametrize('geom', WINO_GEOMS, ids=lambda g: 'x'.join(map(str, g)))
def test_winograd_stride1(geom, tile):
    """AUTO (Winograd) against the fp32 CPU convolution and against the implicit-GEMM kernel of the same library; the
    F(2x2,4x4) and F(3x3,4x4) instances are also pinned explicitly through the PG_TUNE_WINO1_F2 / _F3 bits."""
    from patchgan_amd import engine as E, _lib as L
    from tests.gpu_util import to_view, empty_view, pack, rel_err
    N, Hb, Wb, Ca, Cb, s = geom
    big, small, Wt, Hs, Ws = _mk(*geom)
    bits = {'auto': 0, 'f2': L.TUNE_WINO1_F2, 'f3': L.TUNE_WINO1_F3, 'f3_dma': L.TUNE_WINO1_F3 | L.TUNE_WINO_DMA,
            'f2_dma': L.TUNE_WINO1_F2 | L.TUNE_WINO_DMA}[tile]      # *_dma: the LDS-DMA ring kernels k_wino_gemm_dma
    auto, mfma = E.ConvOp(*geom, L.ALGO_AUTO | bits), E.ConvOp(*geom, 2)
    assert auto.describe(0)[0].startswith('k_wino_gemm') and auto.describe(1)[0].startswith('k_wino_gemm')
    if tile != 'auto':       # F(3x3,4x4) needs Cin % 64 == 0 in each direction, else the F(2x2,4x4) instance runs
        for oc, cin in ((0, Cb), (1, Ca)):
            sym = auto.describe(oc)[0]          # F(3x3,4x4): k_wino_gemm<...,3> or its LDS-DMA form k_wino_gemm_dma<3,...>
            is_f3 = sym.startswith(('k_wino_gemm_dma<3', 'k_wino_gemm_row<', 'k_wino_gemm_row_s3<')) or (sym.startswith('k_wino_gemm<') and sym.endswith(',3>'))
            assert is_f3 == (tile.startswith('f3') and cin % 64 == 0), (oc, sym)
            if tile.endswith('_dma'):             # (the 128-tile-row F(2x2,4x4) instance has no DMA form)
                assert sym.startswith('k_wino_gemm_dma<') or sym == 'k_wino_gemm<2,1,2,2,2,2>', sym
    assert not E.ConvOp(*geom, L.ALGO_AUTO | L.TUNE_WINO_OFF).describe(0)[0].startswith('k_wino')
    assert not mfma.describe(0)[0].startswith('k_wino_gemm')
    bias_a, bias_b = torch.randn(Ca), torch.randn(Cb)
    P = pack(Wt)
    # forward: bias + LeakyReLU fused, strided output view
    want = O.apply_act(F.conv2d(big, Wt, bias_a, stride=1, padding=1), 'leakyrelu')
    outs = []
    for op in (auto, mfma):
        vs = empty_view(N, Hs, Ws, Ca, ld=Ca + 4, off=4)
        op.big2small(to_view(big, ld=Cb + 4, off=0), P, 0, bias_a.cuda(), 0, vs, ACTS['leakyrelu'])
        outs.append(vs.to_nchw())
    torch.cuda.synchronize()
    assert rel_err(outs[0], want) < 1e-5 and rel_err(outs[1], want) < 1e-5 and rel_err(outs[0], outs[1]) < 1e-5
    # data gradient (transposed convolution)
    want = F.conv_transpose2d(small, Wt, bias_b, stride=1, padding=1)
    outs = []
    for op in (auto, mfma):
        vb = empty_view(N, Hb, Wb, Cb, ld=Cb + 8, off=4)
        op.small2big(to_view(small, ld=Ca + 4, off=4), P, 0, bias_b.cuda(), 0, vb)
        outs.append(vb.to_nchw())
    torch.cuda.synchronize()
    assert rel_err(outs[0], want) < 1e-5 and rel_err(outs[1], want) < 1e-5 and rel_err(outs[0], outs[1]) < 1e-5
    # weight (+ bias) gradient: F(4x4, 3x3) on 3x3 tiles of dy where that leaves >= 1024 tiles, else F(4x4, 2x2) (the third geometry)
    from tests.gpu_util import unpack, DEV
    assert auto.describe(2)[0].startswith('k_wino_wgrad_gemm') and not mfma.describe(2)[0].startswith('k_wino')
    t3 = N * ((Hs + 2) // 3) * ((Ws + 2) // 3)
    r = 3 if t3 >= 1024 else 2
    assert auto.kernel_flops(2) == 2.0 * (r + 3) ** 2 * N * ((Hs + r - 1) // r) * ((Ws + r - 1) // r) * Ca * Cb, (r, auto.kernel_flops(2))
    Wr = Wt.clone().requires_grad_(True)
    br = torch.zeros(Ca, requires_grad=True)
    F.conv2d(big, Wr, br, stride=1, padding=1).backward(small)
    outs = []
    for op in (auto, mfma):
        dP = torch.full((16 * Ca * Cb,), float('nan'), device=DEV)
        db = torch.full((Ca,), float('nan'), device=DEV)
        op.wgrad(to_view(small, ld=Ca + 4, off=4), to_view(big, ld=Cb + 8, off=4), dP, 0, db, 0)
        torch.cuda.synchronize()
        assert rel_err(db, br.grad) < 3e-5
        outs.append(unpack(dP, Ca, Cb))
    assert rel_err(outs[0], Wr.grad) < 3e-5 and rel_err(outs[1], Wr.grad) < 3e-5, (rel_err(outs[0], Wr.grad), rel_err(outs[1], Wr.grad))


# stride-2 layers for the polyphase Winograd path: the last two pass the default size heuristic (channel-heavy layers), the
# others are forced onto it with the per-call PG_TUNE_WINO2_ALL bit: odd tile counts, ragged extents, both GEMM tile variants
WINO2_GEOMS = [(4, 32, 32, 128, 64, 2), (2, 64, 64, 64, 32, 2), (3, 36, 44, 96, 40, 2), (16, 16, 16, 256, 128, 2), (9, 64, 64, 160, 64, 2),
               (3, 35, 41, 64, 64, 2), (6, 62, 58, 256, 128, 2), (16, 32, 32, 288, 160, 2)]


@pytest.mark.parametrize('geom', WINO2_GEOMS, ids=lambda g: 'x'.join(map(str, g)))
def test_winograd_stride2_big2small(geom):
    import os
    from patchgan_amd import engine as E, _lib as L
    from tests.gpu_util import to_view, empty_view, pack, rel_err
    tol = 5e-5 if os.environ.get('PATCHGAN_WINO2_TILE') == '4' else 2e-5     # F(4x4,2x2) (opt-in) is 4x less accurate
    N, Hb, Wb, Ca, Cb, s = geom
    big, small, Wt, Hs, Ws = _mk(*geom)
    auto, mfma = E.ConvOp(*geom, L.ALGO_AUTO | L.TUNE_WINO2_ALL), E.ConvOp(*geom, 2)
    assert auto.describe(0)[0].startswith('k_wino_bgemm') and auto.describe(1)[0].startswith('k_wino_bgemm'), auto.describe(0)
    assert not E.ConvOp(*geom, L.ALGO_AUTO | L.TUNE_WINO2_OFF).describe(0)[0].startswith('k_wino')
    bias = torch.randn(Ca)
    P = pack(Wt)
    want = O.apply_act(F.conv2d(big, Wt, bias, stride=2, padding=1), 'leakyrelu')
    outs = []
    for op in (auto, mfma):
        vs = empty_view(N, Hs, Ws, Ca, ld=Ca + 4, off=4)
        op.big2small(to_view(big, ld=Cb + 4, off=0), P, 0, bias.cuda(), 0, vs, ACTS['leakyrelu'])
        outs.append(vs.to_nchw())
    torch.cuda.synchronize()
    assert rel_err(outs[0], want) < tol and rel_err(outs[1], want) < 2e-5, (rel_err(outs[0], want), rel_err(outs[1], want))
    # transposed direction (four parity classes)
    bias_b = torch.randn(Cb)
    want = O.apply_act(F.conv_transpose2d(small, Wt, bias_b, stride=2, padding=1,
                                          output_padding=(Hb - ((Hs - 1) * 2 + 2), Wb - ((Ws - 1) * 2 + 2))), 'tanh')
    outs = []
    for op in (auto, mfma):
        vb = empty_view(N, Hb, Wb, Cb, ld=Cb + 8, off=4)
        op.small2big(to_view(small, ld=Ca + 4, off=4), P, 0, bias_b.cuda(), 0, vb, ACTS['tanh'])
        outs.append(vb.to_nchw())
    torch.cuda.synchronize()
    assert rel_err(outs[0], want) < tol and rel_err(outs[1], want) < 2e-5, (rel_err(outs[0], want), rel_err(outs[1], want))


@pytest.mark.parametrize('geom', [(6, 62, 58, 256, 128, 2), (16, 32, 32, 288, 160, 2), (5, 64, 64, 64, 32, 2), (4, 70, 74, 96, 36, 2),
                                  (9, 64, 64, 512, 128, 2)],
                         ids=lambda g: 'x'.join(map(str, g)))
def test_winograd_stride2_wgrad(geom):
    """Polyphase F(2x2, 3x3) weight gradient, forced wherever the geometry allows by the per-call PG_TUNE_WINO2W_ALL bit."""
    from patchgan_amd import engine as E, _lib as L
    from tests.gpu_util import to_view, unpack, rel_err, DEV
    N, Hb, Wb, Ca, Cb, s = geom
    big, small, Wt, Hs, Ws = _mk(*geom)
    auto, mfma = E.ConvOp(*geom, L.ALGO_AUTO | L.TUNE_WINO2W_ALL), E.ConvOp(*geom, 2)
    assert auto.describe(2)[0].startswith('k_wino_wgrad_gemm'), auto.describe(2)
    assert not E.ConvOp(*geom, L.ALGO_AUTO | L.TUNE_WINO2W_OFF).describe(2)[0].startswith('k_wino')
    Wr = Wt.clone().requires_grad_(True)
    br = torch.zeros(Ca, requires_grad=True)
    F.conv2d(big, Wr, br, stride=2, padding=1).backward(small)
    for op in (auto, mfma):
        dP = torch.full((16 * Ca * Cb,), float('nan'), device=DEV)
        db = torch.full((Ca,), float('nan'), device=DEV)
        op.wgrad(to_view(small, ld=Ca + 4, off=4), to_view(big, ld=Cb + 8, off=4), dP, 0, db, 0)
        torch.cuda.synchronize()
        assert rel_err(db, br.grad) < 3e-5
        assert rel_err(unpack(dP, Ca, Cb), Wr.grad) < 3e-5, rel_err(unpack(dP, Ca, Cb), Wr.grad)


def test_mfma_matches_direct_bitwise_shapes():
    """The two algorithms must agree closely on a cfg2-like layer (enc2 at reduced batch)."""
    from patchgan_amd import engine as E
    from tests.gpu_util import to_view, empty_view, pack, rel_err
    geom = (2, 64, 64, 256, 128, 2)
    big, small, Wt, Hs, Ws = _mk(*geom)
    outs = []
    for algo in (1, 2):
        op = E.ConvOp(*geom, algo)
        vs = empty_view(2, Hs, Ws, 256)
        op.big2small(to_view(big), pack(Wt), 0, None, 0, vs)
        outs.append(vs.to_nchw())
    torch.cuda.synchronize()
    assert rel_err(outs[1], outs[0]) < 1e-5
    want = F.conv2d(big, Wt, None, stride=2, padding=1)
    assert rel_err(outs[1], want) < 1e-5


@pytest.mark.parametrize('shape', [(2, 8, 4, 4), (2, 64, 16, 16), (3, 6, 5, 7), (2, 512, 2, 2), (1, 4, 64, 64),
                                   (2, 64, 64, 64), (2, 6, 50, 47), (1, 136, 48, 48)])
@pytest.mark.parametrize('act', ['leakyrelu', 'relu', 'tanh', 'none'])
def test_instnorm_fwd_bwd(shape, act):
    from patchgan_amd import engine as E
    from tests.gpu_util import to_view, empty_view, rel_err, DEV
    N, C, H, W = shape
    g = torch.Generator().manual_seed(1)
    y = (torch.randn(shape, generator=g) * 2 + 0.5).requires_grad_(True)
    g1 = torch.randn(shape, generator=g)
    g2 = torch.randn(shape, generator=g)
    out = O.apply_act(O.instance_norm(y), act)
    out.backward(g1 + g2)
    vy = to_view(y.detach(), ld=C + 4)
    vo = empty_view(N, H, W, C, ld=C + 8, off=4)
    stats = torch.empty(N * C * 2, device=DEV)
    E.instnorm_act_fwd(vy, vo, stats, E.L.ACT_CODES[act])
    assert rel_err(vo.to_nchw(), out.detach()) < 1e-5
    vdy = empty_view(N, H, W, C)
    E.instnorm_act_bwd(to_view(g1), to_view(g2, ld=C + 4), vy, stats, vdy, E.L.ACT_CODES[act])
    torch.cuda.synchronize()
    assert rel_err(vdy.to_nchw(), y.grad) < 5e-5


def test_instnorm_rejects_single_pixel():
    from patchgan_amd import engine as E
    from tests.gpu_util import to_view, empty_view, DEV
    y = torch.randn(2, 8, 1, 1)
    with pytest.raises(ValueError):
        E.instnorm_act_fwd(to_view(y), empty_view(2, 1, 1, 8), torch.empty(32, device=DEV), 0)


def test_dropout_forward_backward_with_mask():
    from patchgan_amd import engine as E
    from patchgan_amd import _lib as L
    from tests.gpu_util import to_view, empty_view, rel_err, DEV
    shape = (2, 16, 8, 8)
    N, C, H, W = shape
    g = torch.Generator().manual_seed(2)
    y = torch.randn(shape, generator=g).requires_grad_(True)
    g1 = torch.randn(shape, generator=g)
    seed = 0xABCDEF12345
    mask = torch.empty(N * H * W * C, device=DEV)
    L.check(L.load().pg_dropout_mask(mask.data_ptr(), mask.numel(), 0.2, seed, torch.cuda.current_stream().cuda_stream), 'mask')
    m_nchw = mask.view(N, H, W, C).permute(0, 3, 1, 2).cpu()
    keep = m_nchw.mean().item()
    assert 0.7 < keep < 0.9
    out = O.apply_act(O.instance_norm(y), 'leakyrelu') * m_nchw / 0.8
    out.backward(g1)
    vy = to_view(y.detach())
    vo = empty_view(N, H, W, C)
    stats = torch.empty(N * C * 2, device=DEV)
    E.instnorm_act_fwd(vy, vo, stats, 1, 0.2, seed)
    assert rel_err(vo.to_nchw(), out.detach()) < 1e-5
    vdy = empty_view(N, H, W, C)
    E.instnorm_act_bwd(to_view(g1), None, vy, stats, vdy, 1, 0.2, seed)
    torch.cuda.synchronize()
    assert rel_err(vdy.to_nchw(), y.grad) < 5e-5


@pytest.mark.parametrize('act', ['leakyrelu', 'relu', 'tanh', 'sigmoid', 'none'])
@pytest.mark.parametrize('C', [1, 3, 8])
def test_act_bwd(act, C):
    from patchgan_amd import engine as E
    from tests.gpu_util import to_view, empty_view, rel_err
    g = torch.Generator().manual_seed(3)
    y = torch.randn(2, C, 6, 5, generator=g).requires_grad_(True)
    g1 = torch.randn(2, C, 6, 5, generator=g)
    a = O.apply_act(y, act)
    a.backward(g1)
    vdy = empty_view(2, 6, 5, C)
    E.act_bwd(to_view(g1, ld=C + 3, off=1), None, to_view(a.detach()), vdy, E.L.ACT_CODES[act])
    torch.cuda.synchronize()
    assert rel_err(vdy.to_nchw(), y.grad) < 1e-5


@pytest.mark.parametrize('C', [2, 4, 7])
def test_softmax(C):
    from patchgan_amd import engine as E
    from tests.gpu_util import to_view, empty_view, rel_err
    g = torch.Generator().manual_seed(4)
    y = (torch.randn(2, C, 9, 4, generator=g) * 3).requires_grad_(True)
    g1 = torch.randn(2, C, 9, 4, generator=g)
    out = torch.softmax(y, dim=1)
    out.backward(g1)
    vo = empty_view(2, 9, 4, C, ld=C + 1)
    E.softmax_fwd(to_view(y.detach()), vo)
    assert rel_err(vo.to_nchw(), out.detach()) < 1e-6
    vdy = empty_view(2, 9, 4, C)
    E.softmax_bwd(to_view(g1), None, vo, vdy)
    torch.cuda.synchronize()
    assert rel_err(vdy.to_nchw(), y.grad) < 1e-5


@pytest.mark.parametrize('loss_type,C', [('tversky', 1), ('tversky', 4), ('weighted_bce', 1), ('weighted_bce', 3), ('MAE', 2)])
def test_seg_losses(loss_type, C):
    from patchgan_amd import engine as E
    from patchgan_amd.trainer import _LOSS_MODES
    from tests.gpu_util import to_view, empty_view, rel_err, DEV
    g = torch.Generator().manual_seed(5)
    N, H, W = 3, 32, 24
    p = torch.rand(N, C, H, W, generator=g).clamp(1e-4, 1 - 1e-4).requires_grad_(True)
    y = (torch.rand(N, C, H, W, generator=g) > 0.7).float()
    want = O.seg_loss(loss_type, p, y, 200)
    want.backward()
    out = torch.zeros(4, device=DEV)
    vg = empty_view(N, H, W, C)
    E.loss_value_and_grad(to_view(p.detach(), ld=C + 2), to_view(y), 0.0, _LOSS_MODES[loss_type], 200.0, vg, out, 1, N)
    torch.cuda.synchronize()
    assert abs(out[1].item() - want.item()) <= 2e-6 * abs(want.item())
    assert rel_err(vg.to_nchw(), p.grad) < 2e-5


@pytest.mark.parametrize('target', [0.0, 1.0])
def test_bce_const_target(target):
    from patchgan_amd import engine as E
    from patchgan_amd import _lib as L
    from tests.gpu_util import to_view, empty_view, rel_err, DEV
    g = torch.Generator().manual_seed(6)
    p = torch.rand(4, 1, 30, 30, generator=g)
    p[0, 0, 0, 0] = 0.0   # log clamp at -100
    p[1, 0, 0, 0] = 1.0
    p = p.requires_grad_(True)
    want = O.bce(p, torch.full_like(p, target)) * 0.5
    want.backward()
    out = torch.zeros(4, device=DEV)
    vg = empty_view(4, 30, 30, 1)
    E.loss_value_and_grad(to_view(p.detach()), None, target, L.LOSS_BCE, 0.5, vg, out, 0, 4)
    torch.cuda.synchronize()
    assert abs(out[0].item() - want.item()) <= 2e-6 * abs(want.item())
    assert rel_err(vg.to_nchw(), p.grad) < 2e-5


@pytest.mark.parametrize('mode_name,N,C,H,W', [('tversky', 16, 1, 256, 256), ('tversky', 3, 4, 64, 80), ('weighted_bce', 8, 4, 128, 128),
                                               ('weighted_bce', 4, 7, 64, 64), ('MAE', 2, 2, 40, 24), ('bce', 16, 1, 30, 30),
                                               ('bce', 36, 7, 8, 8)])
def test_two_launch_loss_equals_the_staged_one(mode_name, N, C, H, W):
    """pg_loss_reduce_parts + pg_loss_value_grad (one reduction launch, one value + gradient launch: what Trainer.batch runs in one
    process) are bit-identical to the staged pg_loss_reduce (+ combine) / pg_loss_prepare / pg_loss_finalize / pg_loss_grad chain
    that data parallelism keeps (the batch-global terms are all-reduced between its stages) -- with and without a caller-supplied
    gsum2, with and without a gradient, on split and unsplit reductions (four-channel kernel included); beyond N*C = 256 the
    two-launch form refuses and the engine stages."""
    from patchgan_amd import engine as E, _lib as L
    from tests.gpu_util import to_view, empty_view, DEV
    lib = L.load()
    mode = {'tversky': L.LOSS_TVERSKY, 'weighted_bce': L.LOSS_WBCE, 'MAE': L.LOSS_MAE, 'bce': L.LOSS_BCE}[mode_name]
    gmode = {L.LOSS_TVERSKY: 0, L.LOSS_WBCE: 1, L.LOSS_BCE: 1, L.LOSS_MAE: 2}[mode]
    g = torch.Generator().manual_seed(9)
    p = to_view(torch.rand(N, C, H, W, generator=g).clamp(1e-4, 1 - 1e-4), ld=C + 3, off=1)
    y = to_view((torch.rand(N, C, H, W, generator=g) > 0.7).float()) if mode != L.LOSS_BCE else None
    yp, yl = (y.ptr(), y.ld) if y is not None else (None, 0)
    nd = int(lib.pg_loss_reduce_doubles(N, H * W, C))
    # staged
    S = torch.full((nd,), float('nan'), dtype=torch.float64, device=DEV)
    sums = torch.zeros(2, dtype=torch.float64, device=DEV)
    coef = torch.empty(N * C * 2, device=DEV)
    out1, g1 = torch.full((2,), float('nan'), device=DEV), empty_view(N, H, W, C)
    L.check(lib.pg_loss_reduce(p.ptr(), p.ld, yp, yl, 1.0, N, H * W, C, S.data_ptr(), None), 'reduce')
    L.check(lib.pg_loss_prepare(S.data_ptr(), N, C, 0.75, sums.data_ptr(), None), 'prepare')
    L.check(lib.pg_loss_finalize(S.data_ptr(), sums.data_ptr(), mode, N, C, H * W, N, 200.0, 0.75, 0.75, coef.data_ptr(), out1.data_ptr(), None), 'fin')
    L.check(lib.pg_loss_grad(p.ptr(), p.ld, yp, yl, 1.0, coef.data_ptr(), g1.ptr(), g1.ld, N, H * W, C, gmode, None), 'grad')
    # two launches; gsum2 computed inside / supplied; value only
    for supplied in (False, True):
        S2 = torch.full((nd,), float('nan'), dtype=torch.float64, device=DEV)
        Sout = torch.full((N * C * 5,), float('nan'), dtype=torch.float64, device=DEV)
        out2, g2, out3 = torch.full((2,), float('nan'), device=DEV), empty_view(N, H, W, C), torch.full((2,), float('nan'), device=DEV)
        ns = lib.pg_loss_reduce_parts(p.ptr(), p.ld, yp, yl, 1.0, N, H * W, C, S2.data_ptr(), None)
        assert ns >= 1
        gs = sums.data_ptr() if supplied else None
        L.check(lib.pg_loss_value_grad(S2.data_ptr(), ns, Sout.data_ptr(), gs, mode, N, C, H * W, N, 200.0, 0.75, 0.75, p.ptr(), p.ld, yp, yl,
                                       1.0, g2.ptr(), g2.ld, out2.data_ptr(), None), 'value_grad')
        L.check(lib.pg_loss_value_grad(S2.data_ptr(), ns, None, gs, mode, N, C, H * W, N, 200.0, 0.75, 0.75, None, 0, None, 0,
                                       1.0, None, 0, out3.data_ptr(), None), 'value')
        torch.cuda.synchronize()
        assert torch.equal(out2[0], out1[0]) and torch.equal(out3[0], out1[0]), (out1, out2, out3)
        assert torch.equal(g2.to_nchw(), g1.to_nchw())
        assert torch.equal(Sout, S[:N * C * 5])
    if mode_name == 'tversky' and N == 16:
        assert ns > 1                      # the benchmark's segmentation loss runs the split reduction
    big = torch.zeros(300 * 5, dtype=torch.float64, device=DEV)
    assert lib.pg_loss_value_grad(big.data_ptr(), 1, None, None, L.LOSS_BCE, 300, 1, 4, 300, 1.0, 0.75, 0.75, None, 0, None, 0, 1.0, None, 0,
                                  out1.data_ptr(), None) == -1


def test_adam_matches_torch():
    from patchgan_amd import engine as E
    from tests.gpu_util import DEV
    g = torch.Generator().manual_seed(7)
    n = 4099
    p0 = torch.randn(n, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3, betas=(0.9, 0.999))
    p = torch.zeros(n + 1, device=DEV)[:n]
    p.copy_(p0)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for t in range(1, 6):
        grad = torch.randn(n, generator=g) * (10.0 ** (t - 3))
        ref.grad = grad.clone()
        opt.step()
        E.adam_step(p, grad.to(DEV), m, v, t, 1e-3)
    torch.cuda.synchronize()
    assert (p.cpu() - ref.detach()).abs().max().item() < 1e-6


def test_layout_roundtrip():
    from tests.gpu_util import to_view
    x = torch.randn(2, 5, 7, 3)
    v = to_view(x, ld=9, off=2)
    torch.cuda.synchronize()
    assert torch.equal(v.to_nchw().cpu(), x)


@pytest.mark.parametrize('geom,bits', [((6, 62, 58, 256, 128, 2), 0), ((4, 70, 74, 96, 36, 2), 'all'), ((2, 16, 16, 64, 32, 2), 0),
                                       ((16, 32, 32, 288, 160, 2), 0), ((2, 9, 9, 8, 8, 1), 0)],
                         ids=lambda v: 'x'.join(map(str, v)) if isinstance(v, tuple) else str(v))
def test_bwd_big_equals_its_two_halves(geom, bits):
    """pg_conv4x4_bwd_big (ConvTranspose2d backward: weight gradient + data gradient in one call, the transformed dy shared
    where both halves take the polyphase Winograd path) is bit-identical to pg_conv4x4_wgrad + pg_conv4x4_big2small, and within
    the per-kernel tolerance of the CPU convolution's gradients."""
    from patchgan_amd import engine as E, _lib as L
    from tests.gpu_util import to_view, empty_view, pack, unpack, rel_err, DEV
    N, Hb, Wb, Ca, Cb, s = geom
    big, small, Wt, Hs, Ws = _mk(*geom)
    algo = L.ALGO_AUTO | ((L.TUNE_WINO2_ALL | L.TUNE_WINO2W_ALL) if bits == 'all' else 0)
    op = E.ConvOp(*geom, algo)
    P = pack(Wt)
    vs, vb = to_view(small, ld=Ca + 4, off=4), to_view(big, ld=Cb + 8, off=4)
    dP1 = torch.full((16 * Ca * Cb,), float('nan'), device=DEV)
    ds1 = empty_view(N, Hs, Ws, Ca, ld=Ca + 4, off=0)
    op.bwd_big(vs, vb, P, dP1, 0, ds1)
    dP2 = torch.full((16 * Ca * Cb,), float('nan'), device=DEV)
    ds2 = empty_view(N, Hs, Ws, Ca, ld=Ca + 4, off=0)
    op.wgrad(vs, vb, dP2, 0)
    op.big2small(vb, P, 0, None, 0, ds2)
    torch.cuda.synchronize()
    assert torch.equal(dP1, dP2) and torch.equal(ds1.to_nchw(), ds2.to_nchw())
    # against autograd of the transposed convolution y = convT(x; W): dW = wgrad(x, dy), dx = conv(dy, W)
    x = small.clone().requires_grad_(True)
    Wr = Wt.clone().requires_grad_(True)
    F.conv_transpose2d(x, Wr, None, stride=s, padding=1,
                       output_padding=(Hb - ((Hs - 1) * s + 2), Wb - ((Ws - 1) * s + 2))).backward(big)
    assert rel_err(unpack(dP1, Ca, Cb), Wr.grad) < 3e-5 and rel_err(ds1.to_nchw(), x.grad) < 2e-5


@pytest.mark.parametrize('geom', [(4, 32, 32, 128, 64, 2), (3, 36, 44, 64, 32, 2), (16, 16, 16, 256, 128, 2), (2, 64, 64, 64, 32, 2)],
                         ids=lambda g: 'x'.join(map(str, g)))
def test_conv_emits_instancenorm_partials(geom):
    """K5: the polyphase output transforms also write per-sample partial sums / sums of squares of the conv output (fp64, fixed
    order); pg_instnorm_act_fwd_parts normalises from them.  The conv output is bit-identical to the plain call, the merged
    sums equal the sums of the output, and the normalised result equals pg_instnorm_act_fwd's to fp32 rounding."""
    import ctypes
    from patchgan_amd import engine as E, _lib as L
    from tests.gpu_util import to_view, empty_view, pack, rel_err, DEV
    N, Hb, Wb, Ca, Cb, s = geom
    big, small, Wt, Hs, Ws = _mk(*geom)
    op = E.ConvOp(*geom, L.ALGO_AUTO | L.TUNE_WINO2_ALL)
    P = pack(Wt)
    for opcode, src, (Ho, Wo, Co) in ((0, big, (Hs, Ws, Ca)), (1, small, (Hb, Wb, Cb))):
        vin = to_view(src, ld=src.shape[1] + 4, off=0)
        y1, y2 = empty_view(N, Ho, Wo, Co, ld=Co + 4, off=4), empty_view(N, Ho, Wo, Co, ld=Co + 4, off=4)
        chunks = op.stats_chunks(opcode, vin, y1)
        assert chunks > 0, (opcode, op.describe(opcode))
        part = torch.full((N * chunks * Co * 2,), float('nan'), dtype=torch.float64, device=DEV)
        conv = op.big2small if opcode == 0 else op.small2big
        conv(vin, P, 0, None, 0, y1, part=part)
        conv(vin, P, 0, None, 0, y2)
        torch.cuda.synchronize()
        o1 = y1.to_nchw()
        assert torch.equal(o1, y2.to_nchw())
        sums = part.view(N, chunks, Co, 2).sum(1).cpu()
        od = o1.double().cpu()
        assert torch.allclose(sums[..., 0], od.sum((2, 3)), rtol=1e-12, atol=1e-9)
        assert torch.allclose(sums[..., 1], (od * od).sum((2, 3)), rtol=1e-12, atol=1e-9)
        outs, stats = [], []
        for fused in (True, False):
            out = empty_view(N, Ho, Wo, Co, ld=Co + 8, off=4)
            st = torch.empty(N * Co * 2, device=DEV)
            if fused:
                L.check(L.load().pg_instnorm_act_fwd_parts(y1.ptr(), y1.ld, out.ptr(), out.ld, st.data_ptr(), part.data_ptr(), chunks, N,
                                                           Ho * Wo, Co, ACTS['leakyrelu'], 1e-5, 0.0, 0, None), 'parts')
            else:
                E.instnorm_act_fwd(y1, out, st, ACTS['leakyrelu'])
            outs.append(out.to_nchw())
            stats.append(st.clone())
        torch.cuda.synchronize()
        assert rel_err(outs[0], outs[1]) < 1e-6 and rel_err(stats[0], stats[1]) < 1e-6
    # a layer off the polyphase path reports 0 chunks and the *_stats entry point refuses
    plain = E.ConvOp(*geom, L.ALGO_MFMA)
    vin, y = to_view(big, ld=Cb + 4), empty_view(N, Hs, Ws, Ca, ld=Ca + 4, off=4)
    assert plain.stats_chunks(0, vin, y) == 0
    with pytest.raises(RuntimeError):
        plain.big2small(vin, P, 0, None, 0, y, part=torch.empty(16, dtype=torch.float64, device=DEV))


@pytest.mark.parametrize('geom,ld_in', [((4, 64, 64, 64, 3, 2), 4), ((2, 256, 256, 64, 3, 2), 7), ((3, 32, 32, 8, 4, 2), 4), ((2, 64, 64, 4, 3, 2), 3),
                                        ((5, 32, 64, 72, 1, 2), 1), ((2, 256, 256, 128, 2, 2), 2)], ids=lambda v: 'x'.join(map(str, v)) if isinstance(v, tuple) else f'ld{v}')
def test_image_facing_conv_emits_instancenorm_partials(geom, ld_in):
    """K5 on the first encoder layer (unet.py:89-92: Conv2d over the 3-channel image -> InstanceNorm): the persistent image-facing
    kernel k_b2s_tapkp<.., STATS> writes, next to its output, per-sample partial sums / sums of squares (two chunks per 128-pixel
    tile; a lane adds pairs in fp32, everything above that in fp64).  The conv output is bit-identical to the plain call, the merged
    sums equal the sums of the output to 2e-6 (relative to sum |y| resp. sum y^2), and pg_instnorm_act_fwd_parts normalises to the
    same result as the separate statistics pass.  Input views with any pixel stride (x is a slice of the discriminator-input buffer)."""
    from patchgan_amd import engine as E, _lib as L
    from tests.gpu_util import to_view, empty_view, pack, rel_err, DEV
    N, Hb, Wb, Ca, Cb, s = geom
    big, small, Wt, Hs, Ws = _mk(*geom)
    op = E.ConvOp(*geom, L.ALGO_AUTO)
    assert op.describe(0)[0] == f'k_b2s_tapkp<{Cb}>'
    P = pack(Wt)
    vin = to_view(big, ld=ld_in, off=0)
    y1, y2 = empty_view(N, Hs, Ws, Ca, ld=Ca + 4, off=4), empty_view(N, Hs, Ws, Ca, ld=Ca + 4, off=4)
    chunks = op.stats_chunks(0, vin, y1)
    assert chunks == 2 * (Hs * Ws // 128)
    part = torch.full((N * chunks * Ca * 2,), float('nan'), dtype=torch.float64, device=DEV)
    op.big2small(vin, P, 0, None, 0, y1, part=part)
    op.big2small(vin, P, 0, None, 0, y2)
    torch.cuda.synchronize()
    o1 = y1.to_nchw()
    assert torch.equal(o1, y2.to_nchw())
    assert rel_err(o1, F.conv2d(big, Wt, None, stride=2, padding=1)) < 2e-5
    sums = part.view(N, chunks, Ca, 2).sum(1).cpu()
    od = o1.double().cpu()
    assert torch.allclose(sums[..., 0], od.sum((2, 3)), rtol=0, atol=2e-6 * od.abs().sum((2, 3)).max().item())
    assert torch.allclose(sums[..., 1], (od * od).sum((2, 3)), rtol=2e-6, atol=0)
    outs, stats = [], []
    for fused in (True, False):
        out = empty_view(N, Hs, Ws, Ca, ld=Ca + 8, off=4)
        st = torch.empty(N * Ca * 2, device=DEV)
        if fused:
            L.check(L.load().pg_instnorm_act_fwd_parts(y1.ptr(), y1.ld, out.ptr(), out.ld, st.data_ptr(), part.data_ptr(), chunks, N,
                                                       Hs * Ws, Ca, ACTS['leakyrelu'], 1e-5, 0.0, 0, None), 'parts')
        else:
            E.instnorm_act_fwd(y1, out, st, ACTS['leakyrelu'])
        outs.append(out.to_nchw())
        stats.append(st.clone())
    torch.cuda.synchronize()
    assert rel_err(outs[0], outs[1]) < 2e-6 and rel_err(stats[0], stats[1]) < 2e-6
    # with a bias / activation in the epilogue, or on planes that 128-pixel tiles do not divide, there are no partials
    with pytest.raises(RuntimeError):
        op.big2small(vin, P, 0, None, 0, y2, ACTS['leakyrelu'], part=part)
    odd = E.ConvOp(N, 36, 44, Ca, Cb, 2, L.ALGO_AUTO)
    assert odd.stats_chunks(0, to_view(torch.zeros(N, Cb, 36, 44), ld=max(ld_in, Cb)), empty_view(N, 18, 22, Ca)) == 0


@pytest.mark.parametrize('geom,bits', [((6, 62, 58, 256, 128, 2), 0), ((16, 32, 32, 288, 160, 2), 0), ((9, 32, 32, 128, 64, 1), 0),
                                       ((4, 70, 74, 96, 40, 2), 'all'), ((9, 32, 32, 128, 64, 1), 'f3'), ((10, 31, 31, 64, 128, 1), 'f3')],
                         ids=lambda v: 'x'.join(map(str, v)) if isinstance(v, tuple) else str(v))
def test_conv_hand_overs_between_calls(geom, bits):
    """pg_conv_extras: (1) the forward call keeps its polyphase-transformed input (v_keep) and the layer's weight gradient reads
    it (v_pre) instead of transforming again; (2) the transformed weights live in a caller-owned cache (u_cache) and a second
    call marked u_valid skips the weight transform.  Both are bit-identical to the plain calls; a cache marked valid is really
    what the kernel reads (a changed weight tensor has no effect until the cache is refreshed); hand-overs offered to a call
    that does not take such a path are refused."""
    from patchgan_amd import engine as E, _lib as L
    from tests.gpu_util import to_view, empty_view, pack, unpack, DEV
    N, Hb, Wb, Ca, Cb, s = geom
    big, small, Wt, Hs, Ws = _mk(*geom)
    algo = L.ALGO_AUTO | ((L.TUNE_WINO2_ALL | L.TUNE_WINO2W_ALL) if bits == 'all' else L.TUNE_WINO1_F3 if bits == 'f3' else 0)
    op = E.ConvOp(*geom, algo)
    P = pack(Wt)
    vb, vs = to_view(big, ld=Cb + 4, off=0), to_view(small, ld=Ca + 4, off=4)
    # the stride-1 layer shares a transform only between the F(3x3,4x4) forward and the F(4x4,3x3) weight gradient (>= 1024 3x3 tiles)
    shares = s == 2 or (bits == 'f3' and N * ((Hs + 2) // 3) * ((Ws + 2) // 3) >= 1024)

    def fwd(**kw):
        out = empty_view(N, Hs, Ws, Ca, ld=Ca + 4, off=4)
        op.big2small(vb, P, 0, None, 0, out, ACTS['leakyrelu'], **kw)
        return out.to_nchw()

    def dgrad(Pq, **kw):
        out = empty_view(N, Hb, Wb, Cb, ld=Cb + 8, off=4)
        op.small2big(vs, Pq, 0, None, 0, out, **kw)
        return out.to_nchw()

    ref = fwd()
    if shares:                                       # (1) v_keep -> v_pre
        assert op.v_bytes() > 0
        vk = torch.empty(op.v_bytes(), dtype=torch.uint8, device=DEV)
        assert torch.equal(fwd(v_keep=vk), ref)
        dP1 = torch.full((16 * Ca * Cb,), float('nan'), device=DEV)
        dP2 = torch.full((16 * Ca * Cb,), float('nan'), device=DEV)
        op.wgrad(vs, vb, dP1, 0, v_pre=vk)
        op.wgrad(vs, vb, dP2, 0)
        torch.cuda.synchronize()
        assert torch.equal(dP1, dP2)
    else:
        assert op.v_bytes() == 0
    for opcode, call, refv in ((0, lambda **kw: fwd(**kw), ref), (1, lambda **kw: dgrad(P, **kw), dgrad(P))):   # (2) u_cache
        assert op.u_bytes(opcode) > 0
        u = torch.empty(op.u_bytes(opcode), dtype=torch.uint8, device=DEV)
        assert torch.equal(call(u_cache=u, u_valid=False), refv)
        assert torch.equal(call(u_cache=u, u_valid=True), refv)
    u = torch.empty(op.u_bytes(1), dtype=torch.uint8, device=DEV)
    d0 = dgrad(P, u_cache=u, u_valid=False)
    P2 = P * 2
    assert torch.equal(dgrad(P2, u_cache=u, u_valid=True), d0)            # the cache is what is read ...
    assert torch.equal(dgrad(P2, u_cache=u, u_valid=False), dgrad(P2))    # ... until it is refreshed
    # a path without such operands refuses them
    plain = E.ConvOp(*geom, L.ALGO_MFMA)
    assert plain.u_bytes(0) == 0 and plain.v_bytes() == 0
    with pytest.raises(RuntimeError):
        plain.big2small(vb, P, 0, None, 0, empty_view(N, Hs, Ws, Ca, ld=Ca + 4, off=4), u_cache=torch.empty(1 << 20, dtype=torch.uint8, device=DEV))
    with pytest.raises(RuntimeError):
        plain.wgrad(vs, vb, torch.empty(16 * Ca * Cb, device=DEV), 0, v_pre=torch.empty(1 << 20, dtype=torch.uint8, device=DEV))


def _act_grad_from_out(a, name):
    """f'(x) through the activation output a = f(x) (what torch's in-place activations keep)."""
    return 1 - a * a if name == 'tanh' else torch.where(a > 0, torch.ones_like(a), torch.full_like(a, 0.2))


@pytest.mark.parametrize('act', ['tanh', 'leakyrelu'])
@pytest.mark.parametrize('geom,bits,prec', [((9, 32, 32, 128, 64, 1), 0, 'fp32'), ((9, 32, 32, 128, 64, 1), 'f3', 'fp32'),
                                            ((9, 32, 32, 128, 64, 1), 'f3_dma', 'fp32'), ((6, 62, 58, 256, 128, 2), 0, 'fp32'),
                                            ((4, 32, 32, 128, 64, 2), 'w2', 'fp32'), ((2, 64, 64, 128, 64, 2), 0, 'bf16'),
                                            ((2, 31, 31, 128, 256, 1), 0, 'bf16'), ((2, 8, 8, 512, 512, 2), 0, 'bf16'),
                                            ((3, 31, 29, 1, 512, 1), 0, 'fp32'), ((2, 32, 32, 1, 64, 2), 0, 'fp32'), ((3, 31, 29, 1, 512, 1), 0, 'head16')],
                         ids=lambda v: 'x'.join(map(str, v)) if isinstance(v, tuple) else str(v))
def test_activation_backward_in_data_gradient_epilogue(geom, bits, prec, act):
    """pg_conv_extras.mul_t: small2big(dy) * f'(t) in one kernel (the Winograd output transforms, k_wino_gemm's epilogue, the bf16
    LDS-DMA kernel and its split-K reduce) equals small2big followed by pg_act_bwd: bit for bit on fp32 tensors (the same fp32
    product of the same two factors), within one bf16 ulp of the float64 value on bf16 tensors (the fused form rounds once)."""
    from patchgan_amd import engine as E, _lib as L
    from tests.gpu_util import to_view, to_view_bf, empty_view, empty_view_bf, pack, rel_err
    N, Hb, Wb, Ca, Cb, s = geom
    big, small, Wt, Hs, Ws = _mk(*geom)
    tune = {0: 0, 'f3': L.TUNE_WINO1_F3, 'f3_dma': L.TUNE_WINO1_F3 | L.TUNE_WINO_DMA, 'w2': L.TUNE_WINO2_ALL}[bits]
    bf = prec in ('bf16', 'head16')          # head16: fp32 one-channel small -> bf16 big (the discriminator head's data gradient in bf16 mode)
    op = E.ConvOp(*geom, (L.ALGO_BF16 if bf else L.ALGO_AUTO) | tune)
    code = ACTS[act]
    t = O.apply_act(torch.randn(N, Cb, Hb, Wb), act)               # an activation OUTPUT of the layer below
    if bf:
        small, t, Wt = small.bfloat16().float(), t.bfloat16().float(), Wt.bfloat16().float()
    P = pack(Wt)
    mk, em = (to_view_bf, empty_view_bf) if bf else (to_view, empty_view)
    vs, vt = (to_view if prec == 'head16' else mk)(small, ld=Ca + 8, off=8), mk(t, ld=Cb + 8, off=8)
    if Ca == 1:
        assert op.describe(1, op._io(em(N, Hb, Wb, Cb), vs))[0] .startswith('k_s2b_ca1')
    fused, g, ref = em(N, Hb, Wb, Cb, ld=Cb + 8, off=8), em(N, Hb, Wb, Cb), em(N, Hb, Wb, Cb)
    assert op.mul_ok(vs, fused, vt), op.describe(1, op._io(fused, vs))
    op.small2big(vs, P, 0, None, 0, fused, mul=(vt, code))
    op.small2big(vs, P, 0, None, 0, g)
    E.act_bwd(g, None, vt, ref, code)
    torch.cuda.synchronize()
    if not bf:
        if bits == 'f3_dma':      # with a multiplier the register-staged instance runs (one accumulation chain instead of two)
            assert rel_err(fused.to_nchw(), ref.to_nchw()) < 1e-5
        else:
            assert torch.equal(fused.to_nchw(), ref.to_nchw())
        lin = torch.nn.grad.conv2d_input((N, Cb, Hb, Wb), Wt, small, stride=s, padding=1)
        assert rel_err(fused.to_nchw(), lin * _act_grad_from_out(t, act)) < 3e-5
    else:
        lin = torch.nn.grad.conv2d_input((N, Cb, Hb, Wb), Wt.double(), small.double(), stride=s, padding=1)
        want = (lin * _act_grad_from_out(t.double(), act))
        got, wr = fused.to_nchw().double().cpu(), want.float().bfloat16().double()
        assert ((got - wr).abs() <= wr.abs() * 2.0 ** -7 + 1e-5 * want.abs().max()).all(), rel_err(got, wr)
    # a call whose kernel cannot apply the multiplier refuses it
    slow = E.ConvOp(*geom, L.ALGO_DIRECT)
    assert not slow.mul_ok(to_view(small), empty_view(N, Hb, Wb, Cb), to_view(t))
    with pytest.raises(RuntimeError):
        slow.small2big(to_view(small), P, 0, None, 0, empty_view(N, Hb, Wb, Cb), mul=(to_view(t), code))


def test_weight_prep_batch_equals_per_layer_transforms():
    """pg_conv_prep_batch: the Winograd weight transforms (stride-1 F(2x2,4x4) / F(3x3,4x4), both polyphase directions) and the packed
    bf16 weights (plain, 8-channel-pixel form) of several layers made by ONE call are bit-identical to what each convolution call
    writes into its own pg_conv_extras.u_cache; a batch with an item that has no weight preparation is refused as a whole; the data
    gradient inside pg_conv4x4_bwd_big_x reads such a cache."""
    import ctypes
    from patchgan_amd import engine as E, _lib as L
    from tests.gpu_util import to_view, to_view_bf, to_view_bf8, empty_view, empty_view_bf, pack, DEV
    cases = [((6, 62, 58, 256, 128, 2), L.ALGO_AUTO, 0), ((9, 32, 32, 128, 64, 1), L.ALGO_AUTO, 0),
             ((9, 32, 32, 128, 64, 1), L.ALGO_AUTO | L.TUNE_WINO1_F3, 0), ((4, 70, 74, 96, 40, 2), L.ALGO_AUTO | L.TUNE_WINO2_ALL, 0),
             ((2, 64, 64, 128, 64, 2), L.ALGO_BF16, L.IO_MASK), ((2, 31, 31, 128, 256, 1), L.ALGO_BF16, L.IO_MASK),
             ((2, 64, 64, 128, 64, 2), L.ALGO_BF16 | L.TUNE_BF16X_RING, L.IO_MASK), ((2, 64, 64, 64, 4, 2), L.ALGO_BF16, L.IO_MASK)]
    plan, want, flat_parts, off = [], [], [], 0
    for ci, (geom, algo, io) in enumerate(cases):
        N, Hb, Wb, Ca, Cb, s = geom
        big, small, Wt, Hs, Ws = _mk(*geom, seed=ci)
        op = E.ConvOp(*geom, algo)
        P = pack(Wt)
        for opcode in (0, 1):
            if Cb <= 8 and opcode == 1:
                continue                               # (the taps-in-N row GEMM onto a few channels owns no cache)
            nb = op.u_bytes(opcode, io)
            assert nb > 0, (geom, opcode)
            u = torch.zeros(nb, dtype=torch.uint8, device=DEV)
            if io:
                vb = to_view_bf8(big) if Cb <= 8 else to_view_bf(big)
                vs = to_view_bf(small)
                ob, os_ = empty_view_bf(N, Hb, Wb, Cb), empty_view_bf(N, Hs, Ws, Ca)
            else:
                vb, vs = to_view(big, ld=Cb + 4), to_view(small, ld=Ca + 4, off=4)
                ob, os_ = empty_view(N, Hb, Wb, Cb, ld=Cb + 8, off=4), empty_view(N, Hs, Ws, Ca, ld=Ca + 4, off=4)
            if opcode == 0:
                op.big2small(vb, P, 0, None, 0, os_, u_cache=u, u_valid=False)
            else:
                op.small2big(vs, P, 0, None, 0, ob, u_cache=u, u_valid=False)
            plan.append(((ci, opcode, nb), op, opcode, io, off, nb))
            want.append(u)
        flat_parts.append(P)
        off += P.numel()
    flat = torch.cat(flat_parts)
    uc = E.prefill_ucache(plan, flat, flat.device)
    torch.cuda.synchronize()
    assert len(uc) == len(plan)
    for (key, op, opcode, io, p_off, nb), w in zip(plan, want):
        # (the buffers are torch.empty: compare the bytes the transform defines -- the whole buffer but its 256-byte round-up)
        got = uc[key]
        used = nb - 256
        assert torch.equal(got[:used], w[:used]), (key, op.g.key(), opcode)
    # all or nothing: one item without a weight preparation -> PG_EINVAL, nothing launched
    plain = E.ConvOp(6, 62, 58, 256, 128, 2, L.ALGO_MFMA)
    assert plain.u_bytes(0) == 0
    with pytest.raises(RuntimeError):
        E.prefill_ucache(plan[:2] + [(('x',), plain, 0, 0, 0, 1 << 20)], flat, flat.device)
    # pg_conv4x4_bwd_big_x: the data gradient on a prepared cache, bit-identical to the plain call
    geom = cases[0][0]
    N, Hb, Wb, Ca, Cb, s = geom
    big, small, Wt, Hs, Ws = _mk(*geom, seed=0)
    op, P = plan[0][1], flat_parts[0]
    vs, vb = to_view(small, ld=Ca + 4, off=4), to_view(big, ld=Cb + 8, off=4)
    outs = []
    for kw in ({}, {'u_cache': uc[plan[0][0]], 'u_valid': True}):
        dP = torch.full((16 * Ca * Cb,), float('nan'), device=DEV)
        ds = empty_view(N, Hs, Ws, Ca, ld=Ca + 4, off=0)
        op.bwd_big(vs, vb, P, dP, 0, ds, **kw)
        outs.append((dP, ds.to_nchw()))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize('geom', [(9, 32, 32, 128, 64, 1), (16, 32, 32, 512, 256, 1)], ids=lambda g: 'x'.join(map(str, g)))
def test_winograd_stride1_fused_kernel_on_unaligned_outputs(geom):
    """F(3x3,4x4) normally runs as k_wino_gemm_row + k_wino_t_out (16-byte stores); an output view that is only 8-byte aligned takes the
    fully fused nine-accumulator kernel k_wino_gemm<..,3> -- with its input channels split over 2 / 4 workgroups per tile and the slab
    reduce where the grid is small (both geometries in one direction or the other) -- and must give the same convolution."""
    from patchgan_amd import engine as E, _lib as L
    from tests.gpu_util import to_view, empty_view, pack, rel_err
    N, Hb, Wb, Ca, Cb, s = geom
    big, small, Wt, Hs, Ws = _mk(*geom)
    op = E.ConvOp(*geom, L.ALGO_AUTO | L.TUNE_WINO1_F3)
    assert op.describe(0)[0].startswith('k_wino_gemm') and op.describe(1)[0].startswith('k_wino_gemm')
    P = pack(Wt)
    bias_a, bias_b = torch.randn(Ca), torch.randn(Cb)
    outs = []
    for off in (4, 2):                 # 16-byte aligned (row-split pair) / 8-byte aligned (fused kernel)
        vs = empty_view(N, Hs, Ws, Ca, ld=Ca + 4, off=off)
        op.big2small(to_view(big, ld=Cb + 4, off=0), P, 0, bias_a.cuda(), 0, vs, ACTS['leakyrelu'])
        vb = empty_view(N, Hb, Wb, Cb, ld=Cb + 8, off=off)
        op.small2big(to_view(small, ld=Ca + 4, off=4), P, 0, bias_b.cuda(), 0, vb)
        outs.append((vs.to_nchw(), vb.to_nchw()))
    torch.cuda.synchronize()
    want_f = O.apply_act(F.conv2d(big, Wt, bias_a, stride=1, padding=1), 'leakyrelu')
    want_d = F.conv_transpose2d(small, Wt, bias_b, stride=1, padding=1)
    for f, d in outs:
        assert rel_err(f, want_f) < 1e-5 and rel_err(d, want_d) < 1e-5, (rel_err(f, want_f), rel_err(d, want_d))
    assert rel_err(outs[0][0], outs[1][0]) < 1e-5 and rel_err(outs[0][1], outs[1][1]) < 1e-5      # (other association, split-K order)
