"""Helpers for the -m gpu parity tests: NCHW torch tensors <-> engine Views, packed weights."""
import torch

from patchgan_amd import engine as E

DEV = 'cuda'


def to_view(x, ld=None, off=0):
    """NCHW CPU tensor -> NHWC View on the GPU, optionally embedded in a wider buffer (pixel stride ld, channel offset)."""
    N, C, H, W = x.shape
    ld = ld or C
    buf = torch.full((N * H * W * ld + 64,), float('nan'), dtype=torch.float32, device=DEV)
    v = E.View(buf, off, ld, N, H, W, C)
    v.from_nchw(x.to(DEV).float())
    return v


def empty_view(N, H, W, C, ld=None, off=0):
    ld = ld or C
    buf = torch.full((N * H * W * ld + 64,), float('nan'), dtype=torch.float32, device=DEV)
    return E.View(buf, off, ld, N, H, W, C)


def pack(Wt):
    """torch weight [a, b, 4, 4] -> packed P[16][a][b] flat on the GPU."""
    return Wt.permute(2, 3, 0, 1).contiguous().reshape(-1).to(DEV)


def unpack(P, a, b):
    return P.reshape(4, 4, a, b).permute(2, 3, 0, 1).contiguous().cpu()


def rel_err(got, want):
    got, want = got.double().cpu(), want.double().cpu()
    return ((got - want).abs().max() / want.abs().max().clamp_min(1e-30)).item()


def to_view_bf(x, ld=None, off=0):
    """NCHW CPU tensor -> bf16 NHWC View on the GPU (bf16 activation storage), optionally a channel slice of a wider buffer."""
    from patchgan_amd import _lib as L
    N, C, H, W = x.shape
    ld = ld or C
    src = to_view(x)
    buf = torch.full((N * H * W * ld + 64,), float('nan'), dtype=torch.bfloat16, device=DEV)
    v = E.View(buf, off, ld, N, H, W, C, True)
    L.check(L.load().pg_act_fwd_t(src.ptr(), src.ld, v.ptr(), v.ld, v.npix, C, L.ACT_NONE, 0.0, 0, None, 2), 'pg_act_fwd_t')
    return v


def empty_view_bf(N, H, W, C, ld=None, off=0):
    ld = ld or C
    buf = torch.full((N * H * W * ld + 64,), float('nan'), dtype=torch.bfloat16, device=DEV)
    return E.View(buf, off, ld, N, H, W, C, True)


def to_view_bf8(x):
    """NCHW CPU tensor with <= 8 channels -> bf16 NHWC View in 8-channel pixels (ld 8, pad channels zero): the layout the bf16
    kernels take for image-facing tensors."""
    from patchgan_amd import _lib as L
    N, C, H, W = x.shape
    assert C <= 8
    src = to_view(x)
    buf = torch.zeros(N * H * W * 8, dtype=torch.bfloat16, device=DEV)
    v = E.View(buf, 0, 8, N, H, W, C, True)
    L.check(L.load().pg_act_fwd_t(src.ptr(), src.ld, v.ptr(), v.ld, v.npix, C, L.ACT_NONE, 0.0, 0, None, 2), 'pg_act_fwd_t')
    return v
