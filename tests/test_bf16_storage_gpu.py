"""bf16 activation storage: the conv kernels of PG_ALGO_BF16 reading / writing bf16 NHWC tensors (PG_IO_* bits)
and the InstanceNorm / activation kernels with per-tensor storage types (*_t entry points), against the fp32-storage results of
the same kernels fed the bf16-rounded inputs (the arithmetic is identical: only loads widen and stores round) and against the
CPU oracle.  Needs an MI355X."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import patchgan_oracle as O

pytestmark = pytest.mark.gpu

GEOMS = [(2, 64, 64, 128, 64, 2), (3, 34, 38, 40, 36, 2), (2, 32, 32, 64, 32, 1), (2, 4, 4, 512, 64, 2), (1, 8, 8, 160, 96, 2)]


def _mk(N, Hb, Wb, Ca, Cb, s, seed=0):
    g = torch.Generator().manual_seed(seed)
    Hs, Ws = (Hb - 2) // s + 1, (Wb - 2) // s + 1
    big = torch.randn(N, Cb, Hb, Wb, generator=g).bfloat16().float()        # bf16-representable inputs
    small = torch.randn(N, Ca, Hs, Ws, generator=g).bfloat16().float()
    Wt = torch.randn(Ca, Cb, 4, 4, generator=g) / math.sqrt(Cb * 16)
    return big, small, Wt, Hs, Ws


@pytest.mark.parametrize('geom', GEOMS, ids=lambda g: 'x'.join(map(str, g)))
def test_conv_kernels_on_bf16_tensors(geom):
    from patchgan_amd import engine as E, _lib as L
    from tests.gpu_util import to_view, to_view_bf, empty_view, empty_view_bf, pack, unpack, rel_err, DEV
    N, Hb, Wb, Ca, Cb, s = geom
    big, small, Wt, Hs, Ws = _mk(*geom)
    op = E.ConvOp(*geom, L.ALGO_BF16 | L.TUNE_BF16X_OFF)      # (the LDS-DMA kernels: test_lds_dma_bf16_kernels)
    P = pack(Wt)
    bias = torch.randn(Ca)
    # big2small: bf16 in -> {bf16, fp32} out equals the fp32-storage kernel on the same (bf16-representable) input, up to the
    # rounding of the stored output
    ref = empty_view(N, Hs, Ws, Ca, ld=Ca + 4, off=4)
    op.big2small(to_view(big, ld=Cb + 4), P, 0, bias.cuda(), 0, ref, 1)
    for out_bf in (True, False):
        out = (empty_view_bf if out_bf else empty_view)(N, Hs, Ws, Ca, ld=Ca + 4, off=4)
        op.big2small(to_view_bf(big, ld=Cb + 8, off=4), P, 0, bias.cuda(), 0, out, 1)
        torch.cuda.synchronize()
        want = ref.to_nchw().bfloat16().float() if out_bf else ref.to_nchw()
        assert torch.equal(out.to_nchw(), want), (out_bf, rel_err(out.to_nchw(), want))
    assert rel_err(ref.to_nchw(), O.apply_act(F.conv2d(big, Wt, bias, stride=s, padding=1), 'leakyrelu')) < 2e-2
    # small2big
    bias_b = torch.randn(Cb)
    ref = empty_view(N, Hb, Wb, Cb, ld=Cb + 4, off=0)
    op.small2big(to_view(small, ld=Ca + 4, off=4), P, 0, bias_b.cuda(), 0, ref, 3)
    for out_bf in (True, False):
        out = (empty_view_bf if out_bf else empty_view)(N, Hb, Wb, Cb, ld=Cb + 8, off=4)
        op.small2big(to_view_bf(small, ld=Ca + 4, off=4), P, 0, bias_b.cuda(), 0, out, 3)
        torch.cuda.synchronize()
        want = ref.to_nchw().bfloat16().float() if out_bf else ref.to_nchw()
        assert torch.equal(out.to_nchw(), want), (out_bf, rel_err(out.to_nchw(), want))
    # weight gradient: both operands bf16
    dP1 = torch.full((16 * Ca * Cb,), float('nan'), device=DEV)
    dP2 = torch.full((16 * Ca * Cb,), float('nan'), device=DEV)
    op.wgrad(to_view_bf(small, ld=Ca + 4, off=4), to_view_bf(big, ld=Cb + 8, off=4), dP1, 0)
    op.wgrad(to_view(small, ld=Ca + 4, off=4), to_view(big, ld=Cb + 8, off=4), dP2, 0)
    torch.cuda.synchronize()
    assert torch.equal(dP1, dP2)
    Wr = Wt.clone().requires_grad_(True)
    F.conv2d(big, Wr, None, stride=s, padding=1).backward(small)
    assert rel_err(unpack(dP1, Ca, Cb), Wr.grad) < 2e-2
    # ConvTranspose2d backward in one call on bf16 tensors = its two halves
    dP3 = torch.full((16 * Ca * Cb,), float('nan'), device=DEV)
    ds1, ds2 = empty_view_bf(N, Hs, Ws, Ca), empty_view_bf(N, Hs, Ws, Ca)
    vb, vs = to_view_bf(big), to_view_bf(small)
    op.bwd_big(vs, vb, P, dP3, 0, ds1)
    op.big2small(vb, P, 0, None, 0, ds2)
    torch.cuda.synchronize()
    assert torch.equal(dP3, dP1) and torch.equal(ds1.to_nchw(), ds2.to_nchw())
    # tensors the fast bf16 kernels cannot take are refused, not silently misread
    thin = E.ConvOp(2, 16, 16, 16, 3, 2, L.ALGO_BF16)
    with pytest.raises(RuntimeError):
        thin.big2small(empty_view_bf(2, 16, 16, 3, ld=4), torch.zeros(16 * 16 * 3, device=DEV), 0, None, 0, empty_view(2, 8, 8, 16))


XGEOMS = [(2, 64, 64, 128, 64, 2), (2, 32, 32, 64, 64, 2), (1, 24, 20, 96, 128, 2), (2, 31, 29, 64, 64, 1), (2, 8, 8, 512, 512, 2),
          (3, 35, 37, 72, 64, 2), (5, 128, 128, 128, 64, 2), (1, 16, 16, 256, 1024, 2), (2, 2, 2, 128, 128, 2), (2, 31, 31, 128, 256, 1),
          (2, 32, 32, 512, 512, 2), (1, 64, 32, 256, 64, 2), (3, 32, 64, 128, 192, 2), (3, 20, 37, 192, 128, 1)]
# (geometry, direction) pairs the window-staged kernel k_conv_bf16r must take: stride 2, even maps that tile into whole R x 16
# rectangles of (class) pixels, input channels % 64 == 0 -- with a K split (512 -> 512 on 16 x 16), non-square maps, 192 = 1.5 tiles
WIN_EXPECT = {((2, 64, 64, 128, 64, 2), 0), ((2, 64, 64, 128, 64, 2), 1), ((2, 32, 32, 64, 64, 2), 0), ((2, 32, 32, 64, 64, 2), 1),
              ((5, 128, 128, 128, 64, 2), 0), ((5, 128, 128, 128, 64, 2), 1), ((2, 32, 32, 512, 512, 2), 0), ((2, 32, 32, 512, 512, 2), 1),
              ((1, 64, 32, 256, 64, 2), 0), ((1, 64, 32, 256, 64, 2), 1), ((3, 32, 64, 128, 192, 2), 0), ((3, 32, 64, 128, 192, 2), 1)}


@pytest.mark.parametrize('geom', XGEOMS, ids=lambda g: 'x'.join(map(str, g)))
def test_lds_dma_bf16_kernels(geom):
    """k_conv_bf16x (conv_bf16.hip: LDS-DMA operand tiles, 256 x 128 / 128 x 128 / 256 x 64 workgroup tiles, transposed product
    with 16-byte stores) and k_conv_bf16r (window-staged stride-2 form: WIN_EXPECT) on 16-byte-aligned bf16 tensors, both directions, against torch in float64 on the same bf16-representable
    inputs AND weights -- every product is then exact in fp32, so only the summation order differs: fp32 outputs within 2e-5
    (max-norm), bf16 outputs within one bf16 ulp of the rounded reference.  Also: the register-staged kernels (PG_TUNE_BF16X_OFF)
    agree to the same bound, split-K slabs / ragged tiles / odd parity classes / stride 1 are in the list, and a caller-owned
    packed-weight cache marked valid is what the kernel reads."""
    from patchgan_amd import engine as E, _lib as L
    from tests.gpu_util import to_view_bf, empty_view, empty_view_bf, pack, rel_err, DEV
    N, Hb, Wb, Ca, Cb, s = geom
    big, small, Wt, Hs, Ws = _mk(*geom)
    Wt = Wt.bfloat16().float()
    P = pack(Wt)
    op = E.ConvOp(*geom, L.ALGO_BF16 | L.TUNE_BF16X_FLAT)
    ring = E.ConvOp(*geom, L.ALGO_BF16 | L.TUNE_BF16X_RING)
    old = E.ConvOp(*geom, L.ALGO_BF16 | L.TUNE_BF16X_OFF)

    def check(got_view, want, out_bf):
        got = got_view.to_nchw().double().cpu()
        if out_bf:
            ref = want.float().bfloat16().double()
            assert ((got - ref).abs() <= ref.abs() * 2.0 ** -7 + 1e-5 * want.abs().max()).all(), rel_err(got, ref)
        else:
            assert rel_err(got, want) < 2e-5, rel_err(got, want)

    def run(opcode, o, src, out, bias, act, **kw):
        (o.big2small if opcode == 0 else o.small2big)(src, P, 0, bias, 0, out, act, **kw)
        torch.cuda.synchronize()

    for opcode in (0, 1):
        cin, cout = (Cb, Ca) if opcode == 0 else (Ca, Cb)
        if cin % 64:
            continue
        io_in = L.IO_BIG_BF16 if opcode == 0 else L.IO_SMALL_BF16
        name = op.describe(opcode, io_in)[0]
        if (tuple(geom), opcode) in WIN_EXPECT:
            assert name.startswith('k_conv_bf16r'), name
        else:
            assert name.startswith('k_conv_bf16x') and name.endswith(',64>'), name
        assert ring.describe(opcode, io_in)[0].startswith('k_conv_bf16x') and ring.describe(opcode, io_in)[0].endswith(',32>')
        assert not old.describe(opcode, io_in)[0].startswith('k_conv_bf16x')
        bias = torch.randn(cout)
        if opcode == 0:
            lin = F.conv2d(big.double(), Wt.double(), bias.double(), stride=s, padding=1)
            src, oshape = to_view_bf(big, ld=Cb + 8, off=8), (N, Hs, Ws, Ca)
        else:
            lin = torch.nn.grad.conv2d_input((N, Cb, Hb, Wb), Wt.double(), small.double(), stride=s, padding=1) + bias.double().view(1, -1, 1, 1)
            src, oshape = to_view_bf(small, ld=Ca + 16, off=8), (N, Hb, Wb, Cb)
        want = O.apply_act(lin, 'leakyrelu')
        for out_bf in (True, False):
            for o in (op, ring, old):
                out = (empty_view_bf if out_bf else empty_view)(*oshape, ld=cout + 8, off=8)
                run(opcode, o, src, out, bias.cuda(), 1)
                check(out, want, out_bf)
        # caller-owned packed weights: filled by the first call, read (not rebuilt) by a call that marks them valid
        nb = op.u_bytes(opcode, io_in)
        assert nb >= 16 * Ca * Cb * 2
        u = torch.zeros(nb, dtype=torch.uint8, device=DEV)
        out1, out2 = empty_view_bf(*oshape), empty_view_bf(*oshape)
        run(opcode, op, src, out1, None, 0, u_cache=u)
        (op.big2small if opcode == 0 else op.small2big)(src, torch.zeros_like(P), 0, None, 0, out2, 0, u_cache=u, u_valid=True)
        torch.cuda.synchronize()
        check(out1, lin - bias.double().view(1, -1, 1, 1), True)
        assert torch.equal(out1.to_nchw(), out2.to_nchw())
    # weight gradient (k_wgrad_bf16x: [pixel][channel] tiles by LDS-DMA, fragments by transposed LDS reads)
    if Ca % 32 == 0 and Cb % 32 == 0 and Ca >= 64:
        assert op.describe(2, L.IO_BIG_BF16 | L.IO_SMALL_BF16)[0].startswith('k_wgrad_bf16x'), op.describe(2, L.IO_BIG_BF16 | L.IO_SMALL_BF16)
        Wr = Wt.double().clone().requires_grad_(True)
        F.conv2d(big.double(), Wr, None, stride=s, padding=1).backward(small.double())
        from tests.gpu_util import unpack
        pow2 = (Hs & (Hs - 1)) == 0 and (Ws & (Ws - 1)) == 0
        for o in (op, old):
            if o is old and not (pow2 or (Ws >= 16 and Hs >= 2)):
                continue            # the register-staged kernel's pixel decode does not cover this map size
            dP = torch.full((16 * Ca * Cb,), float('nan'), device=DEV)
            o.wgrad(to_view_bf(small, ld=Ca + 16, off=8), to_view_bf(big, ld=Cb + 8, off=8), dP, 0)
            torch.cuda.synchronize()
            assert rel_err(unpack(dP, Ca, Cb), Wr.grad) < 2e-5, (o is op, rel_err(unpack(dP, Ca, Cb), Wr.grad))


@pytest.mark.parametrize('geom', [(16, 128, 128, 128, 64, 2), (8, 128, 128, 256, 128, 2), (2, 64, 64, 128, 64, 2)],
                         ids=lambda g: 'x'.join(map(str, g)))
def test_bf16_conv_emits_instancenorm_partials(geom):
    """K5 on the bf16 kernels: k_conv_bf16x<.., STATS> writes, next to its bf16 output, per-sample partial sums / sums of squares of
    the STORED (bf16-rounded) values, one chunk per workgroup tile; summed over the chunks they equal the sums of the output tensor
    (fp32 in-tile sums: 1e-5 relative to sum |y|), and the InstanceNorm that consumes them equals the one that re-reads y."""
    from patchgan_amd import engine as E, _lib as L
    from tests.gpu_util import to_view_bf, empty_view_bf, pack, rel_err, DEV
    N, Hb, Wb, Ca, Cb, s = geom
    big, small, Wt, Hs, Ws = _mk(*geom)
    P = pack(Wt)
    op = E.ConvOp(*geom, L.ALGO_BF16)
    ran = 0
    for opcode in (0, 1):
        cin, cout = (Cb, Ca) if opcode == 0 else (Ca, Cb)
        if cin % 64:
            continue
        src = to_view_bf(big if opcode == 0 else small, ld=cin + 8, off=8)
        oshape = (N, Hs, Ws, Ca) if opcode == 0 else (N, Hb, Wb, Cb)
        y = empty_view_bf(*oshape, ld=cout + 8, off=8)
        chunks = op.stats_chunks(opcode, src, y)
        if (oshape[1] * oshape[2]) % 256 == 0 and op.describe(opcode, L.IO_MASK)[1] == 1:
            assert chunks > 0, (opcode, op.describe(opcode, L.IO_MASK))
        if not chunks:
            continue
        ran += 1
        part = torch.full((N * chunks * cout * 2,), float('nan'), dtype=torch.float64, device=DEV)
        (op.big2small if opcode == 0 else op.small2big)(src, P, 0, None, 0, y, part=part)
        torch.cuda.synchronize()
        yv = y.to_nchw().double().cpu()
        got = part.view(N, chunks, cout, 2).sum(1).cpu()
        assert ((got[..., 0] - yv.sum((2, 3))).abs() <= 1e-5 * yv.abs().sum((2, 3)) + 1e-6).all()
        assert ((got[..., 1] - (yv * yv).sum((2, 3))).abs() <= 1e-5 * (yv * yv).sum((2, 3)) + 1e-6).all()
        # the same conv without the hand-over writes the same tensor; InstanceNorm from the partials = InstanceNorm from y
        y2 = empty_view_bf(*oshape, ld=cout + 8, off=8)
        (op.big2small if opcode == 0 else op.small2big)(src, P, 0, None, 0, y2)
        torch.cuda.synchronize()
        assert torch.equal(y.to_nchw(), y2.to_nchw())
        o1, o2 = empty_view_bf(*oshape), empty_view_bf(*oshape)
        st1, st2 = torch.empty(N * cout * 2, device=DEV), torch.empty(N * cout * 2, device=DEV)
        E.instnorm_act_fwd(y2, o2, st2, 1)
        L.check(L.load().pg_instnorm_act_fwd_parts_t(y.ptr(), y.ld, o1.ptr(), o1.ld, st1.data_ptr(), part.data_ptr(), chunks, N,
                                                     oshape[1] * oshape[2], cout, 1, 1e-5, 0.0, 0, None, E._dt(y, o1)), 'parts')
        torch.cuda.synchronize()
        assert rel_err(st1.cpu(), st2.cpu()) < 1e-5
        assert rel_err(o1.to_nchw(), o2.to_nchw()) < 2 ** -7
    assert ran == (2 if N >= 8 else 0)          # the small geometry is split along K: no statistics from the kernel, chunks == 0


SEAMS = [(2, 64, 64, 64, 3, 2), (2, 32, 48, 64, 7, 2), (1, 33, 31, 128, 4, 2), (2, 31, 31, 64, 8, 1), (3, 16, 16, 128, 1, 2), (2, 128, 128, 64, 7, 2)]


@pytest.mark.parametrize('geom', SEAMS, ids=lambda g: 'x'.join(map(str, g)))
def test_bf16_kernels_on_image_facing_layers(geom):
    """The few-channel layers next to the images (enc0, d0, the generator head) on the bf16 kernels: the few-channel tensor is a
    bf16 tensor in 8-channel pixels (ld 8, zero pads -- one 16-byte DMA piece per pixel).  big2small = k_conv_bf16x<..,2,64> (the 16
    taps of a pixel are its K = 128 row), weight gradient = k_wgrad_bf16x<..,true> (taps folded into N), small2big = the one-pass
    k_s2b_tapnf<Cb,bf16> (stride 2 onto <= 4 channels) or the bf16 row GEMM k_conv_bf16x<..,3,64> + col2im, with an fp32 result.  Against torch in float64 on bf16-representable operands: 2e-5 for fp32
    results, one bf16 ulp for bf16 results."""
    from patchgan_amd import engine as E, _lib as L
    from tests.gpu_util import to_view_bf, to_view_bf8, empty_view, empty_view_bf, pack, unpack, rel_err, DEV
    N, Hb, Wb, Ca, Cb, s = geom
    big, small, Wt, Hs, Ws = _mk(*geom)
    Wt = Wt.bfloat16().float()
    P = pack(Wt)
    op = E.ConvOp(*geom, L.ALGO_BF16)
    both = L.IO_BIG_BF16 | L.IO_SMALL_BF16
    bias = torch.randn(Ca)
    vb, vs = to_view_bf8(big), to_view_bf(small, ld=Ca + 8, off=8)
    # forward
    assert op.describe(0, L.IO_BIG_BF16)[0].startswith('k_conv_bf16x') and ',2,64>' in op.describe(0, L.IO_BIG_BF16)[0]
    want = O.apply_act(F.conv2d(big.double(), Wt.double(), bias.double(), stride=s, padding=1), 'leakyrelu')
    for out_bf in (True, False):
        out = (empty_view_bf if out_bf else empty_view)(N, Hs, Ws, Ca, ld=Ca + 8, off=8)
        op.big2small(vb, P, 0, bias.cuda(), 0, out, 1)
        torch.cuda.synchronize()
        got = out.to_nchw().double().cpu()
        if out_bf:
            ref = want.float().bfloat16().double()
            assert ((got - ref).abs() <= ref.abs() * 2.0 ** -7 + 1e-5 * want.abs().max()).all(), rel_err(got, ref)
        else:
            assert rel_err(got, want) < 2e-5
    # weight gradient (+ bias gradient from the bf16 dy)
    assert op.describe(2, both)[0].startswith('k_wgrad_bf16x') and op.describe(2, both)[0].endswith(',true>')
    Wr = Wt.double().clone().requires_grad_(True)
    F.conv2d(big.double(), Wr, None, stride=s, padding=1).backward(small.double())
    dP = torch.full((16 * Ca * Cb,), float('nan'), device=DEV)
    db = torch.full((Ca,), float('nan'), device=DEV)
    op.wgrad(vs, vb, dP, 0, db, 0)
    torch.cuda.synchronize()
    assert rel_err(unpack(dP, Ca, Cb), Wr.grad) < 2e-5, rel_err(unpack(dP, Ca, Cb), Wr.grad)
    assert rel_err(db.cpu(), small.double().sum((0, 2, 3))) < 2e-5
    # data gradient / ConvTranspose2d forward onto the few-channel tensor (fp32 result)
    sym = op.describe(1, L.IO_SMALL_BF16)[0]
    if s == 2 and Ca in (32, 64, 128):          # one pass: taps in N on bf16 MFMAs, the D block in LDS (5-8 channels: two launches)
        assert sym == (f'k_s2b_tapnf<{Cb},bf16>' if Cb <= 4 else f'k_s2b_tapnf<4,bf16>+k_s2b_tapnf<{Cb - 4},bf16>'), sym
    else:
        assert 'k_conv_bf16x' in sym and ',3,64>' in sym, sym
    bias_b = torch.randn(Cb)
    lin = torch.nn.grad.conv2d_input((N, Cb, Hb, Wb), Wt.double(), small.double(), stride=s, padding=1) + bias_b.double().view(1, -1, 1, 1)
    out = empty_view(N, Hb, Wb, Cb, ld=Cb + 4, off=4)
    op.small2big(vs, P, 0, bias_b.cuda(), 0, out, 3)
    torch.cuda.synchronize()
    assert rel_err(out.to_nchw(), torch.tanh(lin)) < 2e-5


@pytest.mark.parametrize('shape', [(2, 64, 16, 16), (3, 6, 5, 7), (2, 512, 2, 2), (2, 64, 64, 64), (1, 136, 48, 48)])
@pytest.mark.parametrize('mix', ['all_bf16', 'y32_out16', 'g16_y32_dy32'])
def test_instnorm_act_mixed_storage(shape, mix):
    """InstanceNorm + activation forward / backward with per-tensor storage types: equal to the fp32-storage kernels on the same
    (bf16-representable) inputs, outputs rounded where they are stored as bf16."""
    from patchgan_amd import engine as E
    from tests.gpu_util import to_view, to_view_bf, empty_view, empty_view_bf, rel_err, DEV
    N, C, H, W = shape
    g = torch.Generator().manual_seed(1)
    y = (torch.randn(shape, generator=g) * 2 + 0.5).bfloat16().float()
    g1 = torch.randn(shape, generator=g).bfloat16().float()
    g2 = torch.randn(shape, generator=g).bfloat16().float()
    y_bf = mix == 'all_bf16'
    out_bf = mix in ('all_bf16', 'y32_out16')
    g_bf = mix in ('all_bf16', 'g16_y32_dy32')
    dy_bf = mix == 'all_bf16'
    mk = lambda t, bf, **kw: (to_view_bf if bf else to_view)(t, **kw)
    em = lambda bf, **kw: (empty_view_bf if bf else empty_view)(N, H, W, C, **kw)
    act = 1
    # reference: fp32 storage everywhere
    st0 = torch.empty(N * C * 2, device=DEV)
    o0 = empty_view(N, H, W, C)
    E.instnorm_act_fwd(to_view(y), o0, st0, act)
    d0 = empty_view(N, H, W, C)
    E.instnorm_act_bwd(to_view(g1), to_view(g2), to_view(y), st0, d0, act)
    # mixed storage
    st1 = torch.empty(N * C * 2, device=DEV)
    o1 = em(out_bf, ld=C + 8, off=4)
    vy = mk(y, y_bf, ld=C + 4)
    E.instnorm_act_fwd(vy, o1, st1, act)
    d1 = em(dy_bf, ld=C + 4, off=0)
    E.instnorm_act_bwd(mk(g1, g_bf, ld=C + 4), mk(g2, g_bf), vy, st1, d1, act)
    torch.cuda.synchronize()
    assert torch.equal(st0, st1)
    want_o = o0.to_nchw().bfloat16().float() if out_bf else o0.to_nchw()
    want_d = d0.to_nchw().bfloat16().float() if dy_bf else d0.to_nchw()
    assert torch.equal(o1.to_nchw(), want_o) and torch.equal(d1.to_nchw(), want_d)
    # act_bwd with mixed storage
    a0, a1 = empty_view(N, H, W, C), em(dy_bf)
    E.act_bwd(to_view(g1), to_view(g2), o0, a0, 3)
    E.act_bwd(mk(g1, g_bf), mk(g2, g_bf), o0, a1, 3)
    torch.cuda.synchronize()
    assert torch.equal(a1.to_nchw(), a0.to_nchw().bfloat16().float() if dy_bf else a0.to_nchw())


def test_networks_bf16_storage_vs_fp32_storage(tmp_path):
    """Whole networks at the benchmark width (nf = ndf = 64, 256x256, B = 2): bf16 mode with bf16 activation storage against the
    same bf16 kernels on fp32-stored activations (round-1 behaviour) and against the fp32 CPU oracle.  Stated tolerance: forward
    outputs within 2e-2 (max-norm) of the oracle, two training steps' losses within 5e-2 of the fp32 oracle's and within 2e-2 of the
    fp32-storage run."""
    import numpy as np
    import patchgan_amd as pg
    from tests.golden_util import LOSS_KEYS
    torch.manual_seed(1234)
    g0 = pg.UNet(3, 1, 64, use_dropout=False, activation='leakyrelu', final_act='sigmoid')
    d0 = pg.Discriminator(4, 64, n_layers=3)
    gw = {k: v.clone() for k, v in g0.state_dict().items()}
    dw = {k: v.clone() for k, v in d0.state_dict().items()}
    gen = torch.Generator().manual_seed(7)
    x = torch.rand(2, 3, 256, 256, generator=gen)
    y = (torch.rand(2, 1, 256, 256, generator=gen) > 0.7).float()
    ot = O.OracleTrainer(gw, dw, activation='leakyrelu', final_act='sigmoid', n_layers=3, norm=False, loss_type='tversky')
    with torch.no_grad():
        ref = O.unet_forward(gw, x, 'leakyrelu', 'sigmoid')
        dref = O.disc_forward(dw, torch.cat((x, ref), 1), 3, False)
    want = [ot.batch(x, y, train=True) for _ in range(2)]
    curves = {}
    for storage in (True, False):
        g = pg.UNet(3, 1, 64, use_dropout=False, activation='leakyrelu', final_act='sigmoid')
        d = pg.Discriminator(4, 64, n_layers=3)
        g.load_state_dict(gw)
        d.load_state_dict(dw)
        g.cuda().set_precision('bf16', bf16_storage=storage)
        d.cuda().set_precision('bf16', bf16_storage=storage)
        assert g.engine.act_bf == storage and d.engine.act_bf == storage
        g.train()
        d.train()
        with torch.no_grad():
            out, hid = g(x.cuda(), return_hidden=True)
            dout = d(torch.cat((x.cuda(), out), 1))
        e_out = ((out.cpu() - ref).abs().max() / ref.abs().max()).item()
        e_d = ((dout.cpu() - dref).abs().max() / dref.abs().max()).item()
        assert e_out < 2e-2 and e_d < 2e-2 and torch.isfinite(hid).all(), (storage, e_out, e_d)
        t = pg.Trainer(g, d, str(tmp_path / f's{int(storage)}'))
        t.setup_optimizers(1e-3, 1e-3)
        got = [t.batch(x, y, train=True) for _ in range(2)]
        curves[storage] = np.array([[r[k] for k in LOSS_KEYS] for r in got])
        err = np.abs(curves[storage] - np.array([[r[k] for k in LOSS_KEYS] for r in want])) / np.maximum(np.abs(np.array([[r[k] for k in LOSS_KEYS] for r in want])), 1e-3)
        print(f'bf16 mode, bf16 activation storage {storage}: output {e_out:.1e}, D out {e_d:.1e}, losses vs fp32 oracle per step {err.max(axis=1)}')
        assert err.max() < 5e-2, err
    rel = np.abs(curves[True] - curves[False]) / np.maximum(np.abs(curves[False]), 1e-3)
    assert rel.max() < 2e-2, rel
    # a narrow network keeps fp32 activations (the fast bf16 kernels need >= 32 channels on every interior tensor)
    small = pg.UNet(3, 1, 4).cuda().set_precision('bf16')
    assert small.engine.act_bf is False


@pytest.mark.parametrize('cfg', ['nf32_norm_n5_dropout_softmax', 'nf64_relu_mae_drop', 'nf48_tanh_tversky'])
def test_bf16_storage_other_configurations(cfg, tmp_path):
    """bf16 mode with bf16 activation storage away from the benchmark configuration: nf = ndf = 32 (32-channel layers stay on the
    register-staged bf16 kernels, wider ones take the LDS-DMA kernels: both families in one network), a discriminator with
    InstanceNorm and n_layers = 5 (Conv -> Tanh -> InstanceNorm: no activation-backward fusion there), a 3-class softmax head with
    weighted BCE, ReLU + MAE, and dropout (same counter-based masks as the fp32 mode: compared against the HIP fp32 run of the same
    seed).  Stated tolerance: every loss of two training steps within 5e-2 of the fp32-mode run, generator output within 3e-2
    (measured 1.0e-2 and 2.2e-2)."""
    import numpy as np
    import patchgan_amd as pg
    from tests.golden_util import LOSS_KEYS
    if cfg.startswith('nf32'):
        nf, out_nc, act, fact, loss, nl, norm, drop, B, S = 32, 3, 'leakyrelu', 'softmax', 'weighted_bce', 5, True, True, 2, 256
    elif cfg.startswith('nf48'):     # channel counts 48 .. 384: neither % 64 nor % 32 everywhere, image-facing tensors stay fp32
        nf, out_nc, act, fact, loss, nl, norm, drop, B, S = 48, 2, 'tanh', 'sigmoid', 'tversky', 3, False, False, 1, 256
    else:
        nf, out_nc, act, fact, loss, nl, norm, drop, B, S = 64, 1, 'relu', 'sigmoid', 'MAE', 3, False, True, 1, 256
    torch.manual_seed(99)
    g0 = pg.UNet(3, out_nc, nf, use_dropout=drop, activation=act, final_act=fact)
    d0 = pg.Discriminator(3 + out_nc, nf, n_layers=nl, norm=norm)
    gw = {k: v.clone() for k, v in g0.state_dict().items()}
    dw = {k: v.clone() for k, v in d0.state_dict().items()}
    gen = torch.Generator().manual_seed(3)
    x = torch.rand(B, 3, S, S, generator=gen)
    y = (torch.rand(B, out_nc, S, S, generator=gen) > 0.6).float()
    curves, outs = {}, {}
    for prec in ('fp32', 'bf16'):
        torch.manual_seed(5)                # the dropout seed base
        g = pg.UNet(3, out_nc, nf, use_dropout=drop, activation=act, final_act=fact)
        d = pg.Discriminator(3 + out_nc, nf, n_layers=nl, norm=norm)
        g.load_state_dict(gw)
        d.load_state_dict(dw)
        g.cuda().set_precision(prec)
        d.cuda().set_precision(prec)
        assert g.engine.act_bf == (prec == 'bf16') and d.engine.act_bf == (prec == 'bf16')
        g.eval()
        with torch.no_grad():
            outs[prec] = g(x.cuda()).cpu()
        g.train()
        d.train()
        t = pg.Trainer(g, d, str(tmp_path / prec))
        t.loss_type = loss
        t.setup_optimizers(1e-3, 1e-3)
        curves[prec] = np.array([[r[k] for k in LOSS_KEYS] for r in (t.batch(x, y, train=True) for _ in range(2))])
        assert np.isfinite(curves[prec]).all()
    e_out = ((outs['bf16'] - outs['fp32']).abs().max() / outs['fp32'].abs().max()).item()
    rel = np.abs(curves['bf16'] - curves['fp32']) / np.maximum(np.abs(curves['fp32']), 1e-3)
    print(f'{cfg}: output bf16 vs fp32 mode {e_out:.1e}, losses {rel.max(axis=1)}')
    assert e_out < 3e-2, e_out
    assert rel.max() < 5e-2, rel


@pytest.mark.parametrize('geom', [(2, 32, 32, 128, 64, 2), (2, 32, 32, 128, 128, 2), (1, 17, 19, 64, 64, 1), (2, 18, 21, 128, 128, 1)], ids=lambda g: 'x'.join(map(str, g)))
def test_bf16_conv_epilogue_bodies_every_activation(geom):
    """The bf16 conv kernels dispatch the activation (and the data gradient's multiplier activation) once per tile into a body compiled
    for that code (epi_dispatch in conv_bf16.hip).  Every code through both kernels -- the window-staged one on its two tilings (64 / 128
    output channels) and the flat one (stride 1) -- with a bias, on bf16 and fp32 outputs, against float64: one bf16 ulp of the rounded
    reference / 2e-5; then every multiplier activation with a bf16 t (window kernel: the slice of t through LDS) and an fp32 t."""
    from patchgan_amd import engine as E, _lib as L
    from tests.gpu_util import to_view, to_view_bf, empty_view, empty_view_bf, pack, rel_err
    N, Hb, Wb, Ca, Cb, s = geom
    big, small, Wt, Hs, Ws = _mk(*geom)
    Wt = Wt.bfloat16().float()
    P = pack(Wt)
    op = E.ConvOp(*geom, L.ALGO_BF16)
    names = set()

    def check(got_view, want, out_bf, what):
        got = got_view.to_nchw().double().cpu()
        if out_bf:
            ref = want.float().bfloat16().double()
            assert ((got - ref).abs() <= ref.abs() * 2.0 ** -7 + 1e-5 * want.abs().max()).all(), (what, rel_err(got, ref))
        else:
            assert rel_err(got, want) < 2e-5, (what, rel_err(got, want))

    for opcode in (0, 1):
        cin, cout = (Cb, Ca) if opcode == 0 else (Ca, Cb)
        names.add(op.describe(opcode, L.IO_MASK)[0].split('<')[0])
        bias = torch.randn(cout)
        if opcode == 0:
            lin = F.conv2d(big.double(), Wt.double(), bias.double(), stride=s, padding=1)
            src, oshape = to_view_bf(big), (N, Hs, Ws, Ca)
        else:
            lin = torch.nn.grad.conv2d_input((N, Cb, Hb, Wb), Wt.double(), small.double(), stride=s, padding=1) + bias.double().view(1, -1, 1, 1)
            src, oshape = to_view_bf(small), (N, Hb, Wb, Cb)
        call = op.big2small if opcode == 0 else op.small2big
        for act in ('none', 'leakyrelu', 'relu', 'tanh', 'sigmoid'):
            want = O.apply_act(lin, act)
            for out_bf in (True, False):
                out = (empty_view_bf if out_bf else empty_view)(*oshape, ld=cout + 8, off=8)
                call(src, P, 0, bias.cuda(), 0, out, L.ACT_CODES[act])
                torch.cuda.synchronize()
                check(out, want, out_bf, (opcode, act, out_bf))
        if opcode == 1:
            lin0 = lin - bias.double().view(1, -1, 1, 1)
            for mact in ('none', 'leakyrelu', 'relu', 'tanh', 'sigmoid'):
                t64 = O.apply_act(torch.randn(N, Cb, Hb, Wb).double(), mact).float().bfloat16().double()     # an activation OUTPUT, bf16-representable
                d = {'none': torch.ones_like(t64), 'leakyrelu': torch.where(t64 > 0, 1.0, 0.2).double(), 'relu': (t64 > 0).double(),
                     'tanh': 1 - t64 * t64, 'sigmoid': t64 * (1 - t64)}[mact]
                for out_bf in (True, False):
                    tv = (to_view_bf if out_bf else to_view)(t64.float(), ld=Cb + 8, off=8)
                    out = (empty_view_bf if out_bf else empty_view)(*oshape, ld=cout + 8, off=8)
                    if not op.mul_ok(src, out, tv):
                        continue
                    op.small2big(src, P, 0, None, 0, out, mul=(tv, L.ACT_CODES[mact]))
                    torch.cuda.synchronize()
                    check(out, lin0 * d, out_bf, ('mul', mact, out_bf))
    assert names <= {'k_conv_bf16r', 'k_conv_bf16x'} and (s == 1 or 'k_conv_bf16r' in names), names
