"""Per-layer parity of the kernels the benchmark really times: every conv call of one G+D step of bench.py's configurations
(tests/bench_layers.py: cfg2 at batch 16 / 32, cfg4 at batch 8 / 16; fp32 tensors under the default planner and bf16 activation
storage) is run through the C ABI on random operands, with the operand storage the engines use (skip-buffer slices with ld = 2 C,
8-channel bf16 pixels), and compared with torch float64 ON THE GPU (seconds per layer, no CPU oracle).

Stated tolerances (relative max-norm, as the per-kernel tests): fp32 results 2e-5 (forward / data gradient) and 3e-5 (weight
gradient); bf16 mode, operands AND weights bf16-representable so that every product is exact in fp32: fp32 results 2e-5 (3e-5
weight gradient), bf16 results within one bf16 ulp of the rounded float64 value.
Each case's id carries the kernel symbol bench.py's conv_kernels table shows for it, and the test asserts that the planner of the
loaded library still picks exactly that symbol (tests/golden/bench_kernel_plan.json): a planner change cannot silently un-test a
kernel.  The hand-overs the engines use on those layers ride along: InstanceNorm partials from the conv epilogue, the V hand-over
forward -> weight gradient, the per-step transformed / packed weight cache, the activation backward in the data-gradient
epilogue, and the one-call ConvTranspose2d backward.  Needs an MI355X."""
import math

import pytest
import torch
import torch.nn.functional as F

from tests import bench_layers as BL

pytestmark = pytest.mark.gpu

_CASES = BL.all_cases()
_PLAN = BL.committed_plan()
_ACT_SLOPE = 0.2


def _act64(x, name):
    if name == 'none':
        return x
    if name == 'leakyrelu':
        return torch.where(x > 0, x, x * _ACT_SLOPE)
    if name == 'tanh':
        return torch.tanh(x)
    if name == 'sigmoid':
        return torch.sigmoid(x)
    if name == 'relu':
        return torch.relu(x)
    raise ValueError(name)


def _dact_from_out(a, name):
    """f'(x) through the activation output a = f(x)."""
    if name == 'tanh':
        return 1 - a * a
    if name == 'leakyrelu':
        return torch.where(a > 0, torch.ones_like(a), torch.full_like(a, _ACT_SLOPE))
    raise ValueError(name)


class _Truth:
    """Random operands of one (geometry, mode) in float64 NCHW on the GPU (bf16-representable in bf16 mode) and, lazily, the
    float64 reference of each op.  One instance is alive at a time (the cases of a layer are consecutive)."""

    def __init__(self, geom, bfm):
        N, Hb, Wb, Ca, Cb, s = geom
        self.geom, self.bfm = geom, bfm
        self.Hs, self.Ws = (Hb - 2) // s + 1, (Wb - 2) // s + 1
        g = torch.Generator(device='cuda').manual_seed(hash(geom) & 0xFFFFFF)

        def rnd(*shape, scale=1.0):
            x = torch.randn(*shape, device='cuda', generator=g) * scale
            return (x.bfloat16().float() if bfm else x).double()
        self.big = rnd(N, Cb, Hb, Wb)
        self.small = rnd(N, Ca, self.Hs, self.Ws)
        self.W = rnd(Ca, Cb, 4, 4, scale=1.0 / math.sqrt(Cb * 16))
        self.bias_a, self.bias_b = rnd(Ca), rnd(Cb)
        self.t_big = torch.tanh(rnd(N, Cb, Hb, Wb))          # an activation OUTPUT with big's shape (for the epilogue multiplier)
        if bfm:
            self.t_big = self.t_big.float().bfloat16().double()
        self._ref = {}

    def ref(self, kind):
        if kind not in self._ref:
            N, Hb, Wb, Ca, Cb, s = self.geom
            if kind == 'b2s':
                r = F.conv2d(self.big, self.W, None, stride=s, padding=1)
            elif kind == 's2b':
                opad = (Hb - ((self.Hs - 1) * s + 2), Wb - ((self.Ws - 1) * s + 2))
                r = F.conv_transpose2d(self.small, self.W, None, stride=s, padding=1, output_padding=opad)
                assert tuple(r.shape[2:]) == (Hb, Wb)
            elif kind == 'wgrad':
                r = torch.nn.grad.conv2d_weight(self.big, self.W.shape, self.small, stride=s, padding=1)
            self._ref[kind] = r
        return self._ref[kind]


_truth = [None]


def _get_truth(geom, bfm):
    t = _truth[0]
    if t is None or t.geom != geom or t.bfm != bfm:
        _truth[0] = None
        torch.cuda.empty_cache()
        t = _truth[0] = _Truth(geom, bfm)
    return t


def _store(x64, spec, fill=float('nan')):
    """float64 NCHW tensor -> engine View with the storage `spec` describes (NHWC; the slice sits at the END of a buffer ldm times
    as wide, the rest of which is NaN; 8-channel bf16 pixels are zero padded)."""
    from patchgan_amd import engine as E
    N, C, H, W = x64.shape
    if spec.pad8:
        ld, off, fill = 8, 0, 0.0
    elif spec.ld is not None:
        ld, off = spec.ld, spec.off
    else:
        ld, off = C * spec.ldm, C * (spec.ldm - 1)
    dt = torch.bfloat16 if spec.bf else torch.float32
    buf = torch.full((N * H * W * ld + 64,), fill, dtype=dt, device='cuda')
    buf[:N * H * W * ld].view(N, H, W, ld)[..., off:off + C] = x64.permute(0, 2, 3, 1).to(dt)
    return E.View(buf, off, ld, N, H, W, C, spec.bf)


def _empty(N, H, W, C, spec):
    return _store(torch.zeros(N, C, H, W, dtype=torch.float64, device='cuda') * float('nan'), spec)


def _read(v):
    n = v.N * v.H * v.W * v.ld          # (views made by _store: off < ld)
    return v.t[:n].view(v.N, v.H, v.W, v.ld)[..., v.off:v.off + v.C].permute(0, 3, 1, 2).double()


def _check(got64, want64, out_bf, tol, what, lin64=None):
    """lin64: the pre-activation values where an activation with |f'| <= 1 follows (tanh / sigmoid squash the output range to 1
    while the kernel's absolute error is that of the convolution sum: the bound is relative to the larger of the two scales)."""
    scale = want64.abs().max().item()
    if lin64 is not None:
        scale = max(scale, lin64.abs().max().item())
    assert math.isfinite(scale) and scale > 0, what
    if out_bf:
        ref = want64.float().bfloat16().double()
        bad = (got64 - ref).abs() > ref.abs() * 2.0 ** -7 + 1e-5 * scale
        assert not bad.any().item(), (what, 'beyond one bf16 ulp', ((got64 - ref).abs().max() / scale).item())
    else:
        err = ((got64 - want64).abs().max() / scale).item()
        _FP32_ERR.append((err, tol, what))
        assert err < tol, (what, err, tol)
    return scale


_FP32_ERR = []          # (relative max-norm error vs float64, its bound, call) of every fp32 check of this session


def _unpack(dP, Ca, Cb):
    return dP.view(4, 4, Ca, Cb).permute(2, 3, 0, 1).double()


def _ids(cs):
    return cs.key + '-' + '+'.join(_PLAN.get(cs.key, ['?']))


@pytest.mark.parametrize('cs', _CASES, ids=_ids)
def test_bench_layer_call_vs_float64(cs):
    from patchgan_amd import engine as E, _lib as L
    N, Hb, Wb, Ca, Cb, s = cs.geom
    bfm = cs.mode == 'bf16'
    op = cs.convop()
    # the kernel this case pins is the one the committed plan (and bench.py's conv_kernels table) names
    assert cs.symbols() == _PLAN[cs.key], (cs.key, cs.symbols(), _PLAN[cs.key])
    T = _get_truth(cs.geom, bfm)
    Hs, Ws = T.Hs, T.Ws
    P = T.W.permute(2, 3, 0, 1).contiguous().reshape(-1).float()
    tol_f, tol_w = 2e-5, 3e-5
    sync = torch.cuda.synchronize

    if cs.op == 'b2s':
        src = _store(T.big, cs.big)
        bias = T.bias_a.float() if cs.bias else None
        lin = T.ref('b2s') + (T.bias_a.view(1, -1, 1, 1) if cs.bias else 0)
        want = _act64(lin, cs.act)
        out = _empty(N, Hs, Ws, Ca, cs.small)
        op.big2small(src, P, 0, bias, 0, out, L.ACT_CODES[cs.act])
        sync()
        ref_out = _read(out)
        _check(ref_out, want, cs.small.bf, tol_f, cs.key, lin)
        plain = ref_out
        # hand-overs the engine uses on this layer: all bit-identical to the plain call
        if not cs.bias and cs.act == 'none':
            chunks = op.stats_chunks(0, src, out)
            if chunks:                      # K5: InstanceNorm partial sums from the conv epilogue
                part = torch.full((N * chunks * Ca * 2,), float('nan'), dtype=torch.float64, device='cuda')
                o2 = _empty(N, Hs, Ws, Ca, cs.small)
                op.big2small(src, P, 0, None, 0, o2, part=part)
                sync()
                got2 = _read(o2)
                assert torch.equal(got2, plain), cs.key
                sums = part.view(N, chunks, Ca, 2).sum(1)
                # (the bf16 epilogue sums in fp32 inside a tile; the image-facing fp32 kernel adds PAIRS in fp32, then fp64)
                tol_s = 1e-5 if bfm else 2e-6 if Cb <= 4 else 1e-12
                assert torch.allclose(sums[..., 0], got2.sum((2, 3)), rtol=tol_s, atol=tol_s * got2.abs().sum((2, 3)).max().item())
                assert torch.allclose(sums[..., 1], (got2 * got2).sum((2, 3)), rtol=max(tol_s, 1e-9), atol=1e-9)
        io = cs.io
        nb = op.u_bytes(0, io) if E.ConvOp.fits(src, out) and (io or E.ConvOp._aligned(src, out)) else 0
        if nb:                              # per-step weight cache: filled by one call, read by the next
            u = torch.zeros(nb, dtype=torch.uint8, device='cuda')
            for valid in (False, True):
                o3 = _empty(N, Hs, Ws, Ca, cs.small)
                op.big2small(src, P if not valid else torch.zeros_like(P), 0, bias, 0, o3, L.ACT_CODES[cs.act], u_cache=u, u_valid=valid)
                sync()
                assert torch.equal(_read(o3), plain), (cs.key, 'u_cache', valid)

    elif cs.op == 's2b':
        src = _store(T.small, cs.small)
        bias = T.bias_b.float() if cs.bias else None
        lin = T.ref('s2b') + (T.bias_b.view(1, -1, 1, 1) if cs.bias else 0)
        want = _act64(lin, cs.act)
        out = _empty(N, Hb, Wb, Cb, cs.big)
        op.small2big(src, P, 0, bias, 0, out, L.ACT_CODES[cs.act])
        sync()
        plain = _read(out)
        _check(plain, want, cs.big.bf, tol_f, cs.key, lin)
        if not cs.bias and cs.act == 'none' and cs.role == 'fwd':
            chunks = op.stats_chunks(1, src, out)
            if chunks:
                part = torch.full((N * chunks * Cb * 2,), float('nan'), dtype=torch.float64, device='cuda')
                o2 = _empty(N, Hb, Wb, Cb, cs.big)
                op.small2big(src, P, 0, None, 0, o2, part=part)
                sync()
                got2 = _read(o2)
                assert torch.equal(got2, plain), cs.key
                sums = part.view(N, chunks, Cb, 2).sum(1)
                tol_s = 1e-5 if bfm else 1e-12
                assert torch.allclose(sums[..., 0], got2.sum((2, 3)), rtol=tol_s, atol=tol_s * got2.abs().sum((2, 3)).max().item())
        if cs.role == 'dgrad' and cs.layer.startswith('d') and not cs.layer.startswith('dec') and not cs.layer.startswith('d0'):
            # the activation backward of the layer below (LeakyReLU under d1, Tanh otherwise) in this kernel's epilogue
            below = 'leakyrelu' if cs.layer.startswith('d1') else 'tanh'
            t64 = T.t_big if below == 'tanh' else _act64(T.t_big * 3 - 1, 'leakyrelu')
            if bfm:
                t64 = t64.float().bfloat16().double()
            tv = _store(t64, cs.big)
            o4 = _empty(N, Hb, Wb, Cb, cs.big)
            if op.mul_ok(src, o4, tv):
                op.small2big(src, P, 0, None, 0, o4, mul=(tv, L.ACT_CODES[below]))
                sync()
                _check(_read(o4), T.ref('s2b') * _dact_from_out(t64, below), cs.big.bf, 3e-5, cs.key + ' x f\'(t)')
        io = cs.io
        nb = op.u_bytes(1, io) if E.ConvOp.fits(src, out) and (io or E.ConvOp._aligned(src, out)) else 0
        if nb:
            u = torch.zeros(nb, dtype=torch.uint8, device='cuda')
            for valid in (False, True):
                o3 = _empty(N, Hb, Wb, Cb, cs.big)
                op.small2big(src, P if not valid else torch.zeros_like(P), 0, bias, 0, o3, L.ACT_CODES[cs.act], u_cache=u, u_valid=valid)
                sync()
                assert torch.equal(_read(o3), plain), (cs.key, 'u_cache', valid)

    elif cs.op == 'wgrad':
        vs, vb = _store(T.small, cs.small), _store(T.big, cs.big)
        want = T.ref('wgrad')
        dP = torch.full((16 * Ca * Cb,), float('nan'), device='cuda')
        db = torch.full((Ca + 4,), float('nan'), device='cuda') if cs.bias else None
        op.wgrad(vs, vb, dP, 0, db, 0)
        sync()
        _check(_unpack(dP, Ca, Cb), want, False, tol_w, cs.key)
        if cs.bias:
            _check(db[:Ca].double(), T.small.sum((0, 2, 3)), False, tol_w, cs.key + ' dbias')
        if not bfm and op.v_bytes() and E.ConvOp._aligned(vs, vb):       # the forward call's transformed input, handed over
            vk = torch.empty(op.v_bytes(), dtype=torch.uint8, device='cuda')
            y = _empty(N, Hs, Ws, Ca, cs.small)
            op.big2small(vb, P, 0, None, 0, y, v_keep=vk)
            dP2 = torch.full((16 * Ca * Cb,), float('nan'), device='cuda')
            op.wgrad(vs, vb, dP2, 0, v_pre=vk)
            sync()
            assert torch.equal(dP2, dP), (cs.key, 'v_keep -> v_pre')

    else:   # bwd_big: ConvTranspose2d backward in one call: dW = wgrad(x = small, dy = big), dx = conv(dy, W)
        vs, vb = _store(T.small, cs.small), _store(T.big, cs.big)
        dP = torch.full((16 * Ca * Cb,), float('nan'), device='cuda')
        ds = _empty(N, Hs, Ws, Ca, BL.Operand(cs.small.bf))
        op.bwd_big(vs, vb, P, dP, 0, ds)
        sync()
        _check(_unpack(dP, Ca, Cb), T.ref('wgrad'), False, tol_w, cs.key + ' dW')
        _check(_read(ds), T.ref('b2s'), cs.small.bf, tol_f, cs.key + ' dx')


def test_zz_report_fp32_maxima():
    """Runs last in this file: prints the measured maxima behind DESIGN.md section 4's tolerance paragraph (forward / data gradient bound
    2e-5, weight gradient 3e-5; SURVEY 8(d)'s forward gate is 1e-5), split by kernel family."""
    if not _FP32_ERR:
        pytest.skip('no fp32 check ran in this session')
    for bound in sorted({b for _, b, _ in _FP32_ERR}):
        rows = [(e, w + ' ' + '+'.join(_PLAN.get(w.split(' ')[0], ['?']))) for e, b, w in _FP32_ERR if b == bound]
        wino = [(e, w) for e, w in rows if 'wino' in w]
        rest = [(e, w) for e, w in rows if 'wino' not in w]
        for name, part in (('Winograd kernels', wino), ('other kernels', rest)):
            if part:
                e, w = max(part)
                print(f'fp32 bench-layer checks, bound {bound:.0e}, {name}: {len(part)} checks, max error {e:.2e} ({w})')
    assert max(e / b for e, b, _ in _FP32_ERR) < 1.0
