"""ctypes binding of libpatchgan_hip.so (the C ABI declared in include/patchgan_hip.h).

There is NO fallback: if the shared library is missing or a call fails, this module raises.  The
library is built in-tree by ``__graft_entry__.build()`` / ``make -C patchgan_amd/csrc``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PATCHGAN_LIB points at an alternative build of the same C ABI (A/B timing of kernel variants on one box)
LIB_PATH = os.environ.get('PATCHGAN_LIB') or os.path.join(_HERE, 'libpatchgan_hip.so')

ACT_NONE, ACT_LEAKY, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4
ACT_CODES = {None: ACT_NONE, 'none': ACT_NONE, 'leakyrelu': ACT_LEAKY, 'relu': ACT_RELU, 'tanh': ACT_TANH,
             'sigmoid': ACT_SIGMOID}
ALGO_AUTO, ALGO_DIRECT, ALGO_MFMA, ALGO_BF16 = 0, 1, 2, 3
ALGO_MASK = 0xF
IO_BIG_BF16, IO_SMALL_BF16, IO_MASK = 0x10000, 0x20000, 0x30000      # bf16 activation storage (PG_IO_*)
# per-call tuning bits OR-ed into `algo` (include/patchgan_hip.h PG_TUNE_*)
TUNE_WINO2_ALL, TUNE_WINO2_OFF, TUNE_WINO2W_ALL, TUNE_WINO2W_OFF = 0x010, 0x020, 0x040, 0x080
TUNE_WINO_OFF, TUNE_WINOW_OFF, TUNE_WINO1_F2, TUNE_WINO1_F3, TUNE_WINO_DMA = 0x100, 0x200, 0x400, 0x800, 0x1000
TUNE_BF16X_OFF, TUNE_BF16X_RING, TUNE_BF16X_FLAT = 0x2000, 0x4000, 0x8000
TUNE_S3_OFF = 0x40000        # the three Winograd GEMM families on the fp32 MFMA instead of their split-bf16 forms k_*_s3 (PG_TUNE_S3_OFF)
LOSS_TVERSKY, LOSS_WBCE, LOSS_MAE, LOSS_BCE = 0, 1, 2, 3
OP_BIG2SMALL, OP_SMALL2BIG, OP_WGRAD = 0, 1, 2

_ERR = {-1: 'PG_EINVAL (bad shape / null or misaligned pointer)', -2: 'PG_EWORKSPACE (workspace too small)',
        -3: 'PG_ELAUNCH (HIP launch failed)'}


class ConvGeom(ctypes.Structure):
    """struct pg_conv_geom"""
    _fields_ = [('N', ctypes.c_int), ('Hb', ctypes.c_int), ('Wb', ctypes.c_int), ('Hs', ctypes.c_int),
                ('Ws', ctypes.c_int), ('Ca', ctypes.c_int), ('Cb', ctypes.c_int), ('stride', ctypes.c_int)]

    def key(self):
        return (self.N, self.Hb, self.Wb, self.Hs, self.Ws, self.Ca, self.Cb, self.stride)


_p = ctypes.c_void_p
_i = ctypes.c_int
_l = ctypes.c_long
_f = ctypes.c_float
_sz = ctypes.c_size_t
_u64 = ctypes.c_uint64
_G = ctypes.POINTER(ConvGeom)


class ConvExtras(ctypes.Structure):
    """struct pg_conv_extras: optional hand-overs between calls on the same layer"""
    _fields_ = [('part', ctypes.c_void_p), ('v_keep', ctypes.c_void_p), ('v_pre', ctypes.c_void_p), ('u_cache', ctypes.c_void_p),
                ('u_valid', ctypes.c_int), ('mul_t', ctypes.c_void_p), ('mul_ld', ctypes.c_int), ('mul_act', ctypes.c_int)]


_X = ctypes.POINTER(ConvExtras)


class ConvPrepItem(ctypes.Structure):
    """struct pg_conv_prep_item: one layer / direction of pg_conv_prep_batch"""
    _fields_ = [('g', ConvGeom), ('op', ctypes.c_int), ('algo', ctypes.c_int), ('ws_bytes', ctypes.c_size_t), ('P', ctypes.c_void_p),
                ('u', ctypes.c_void_p)]


# name -> (restype, argtypes); mirrors include/patchgan_hip.h one to one
SIGNATURES = {
    'pg_version': (_i, []),
    'pg_conv_max_tensor_bytes': (_sz, []),
    'pg_conv_workspace_bytes': (_sz, [_G, _i]),
    'pg_conv_describe': (_i, [_G, _i, _sz, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_l)]),
    'pg_conv_mul_ok': (_i, [_G, _i, _sz]),
    'pg_conv_kernel': (_i, [_G, _i, _sz, ctypes.c_char_p, _sz, ctypes.POINTER(_i), ctypes.POINTER(ctypes.c_double)]),
    'pg_conv_kernel_flops': (_i, [_G, _i, _sz, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    'pg_conv_time_next': (_i, [_p, _p]),
    'pg_conv_time_next2': (_i, [_p, _p, _p, _p]),
    'pg_conv4x4_bwd_big': (_i, [_p, _i, _p, _i, _p, _p, _p, _i, _G, _i, _p, _sz, _p]),
    'pg_conv4x4_bwd_big_x': (_i, [_p, _i, _p, _i, _p, _p, _p, _i, _G, _i, _p, _sz, _p, _X]),
    'pg_conv4x4_big2small': (_i, [_p, _i, _p, _p, _p, _i, _G, _i, _i, _p, _sz, _p]),
    'pg_conv4x4_small2big': (_i, [_p, _i, _p, _p, _p, _i, _G, _i, _i, _p, _sz, _p]),
    'pg_conv4x4_wgrad': (_i, [_p, _i, _p, _i, _p, _p, _G, _i, _p, _sz, _p]),
    'pg_conv_stats_chunks': (_i, [_G, _i, _i, _sz]),
    'pg_conv_u_bytes': (_sz, [_G, _i, _i, _sz]),
    'pg_conv_v_bytes': (_sz, [_G, _i, _sz]),
    'pg_conv4x4_big2small_x': (_i, [_p, _i, _p, _p, _p, _i, _G, _i, _i, _p, _sz, _p, _X]),
    'pg_conv4x4_small2big_x': (_i, [_p, _i, _p, _p, _p, _i, _G, _i, _i, _p, _sz, _p, _X]),
    'pg_conv4x4_wgrad_x': (_i, [_p, _i, _p, _i, _p, _p, _G, _i, _p, _sz, _p, _X]),
    'pg_conv_prep_batch': (_i, [_i, ctypes.POINTER(ConvPrepItem), _p]),
    'pg_instnorm_act_fwd_parts': (_i, [_p, _i, _p, _i, _p, _p, _i, _i, _i, _i, _i, _f, _f, _u64, _p]),
    'pg_instnorm_workspace_bytes': (_sz, [_i, _i, _i]),
    'pg_instnorm_act_fwd': (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _f, _f, _u64, _p, _sz, _p]),
    'pg_instnorm_act_bwd': (_i, [_p, _i, _p, _i, _p, _i, _p, _p, _i, _i, _i, _i, _i, _f, _u64, _p, _sz, _p]),
    'pg_act_bwd': (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _l, _i, _i, _f, _u64, _p]),
    'pg_act_fwd': (_i, [_p, _i, _p, _i, _l, _i, _i, _f, _u64, _p]),
    'pg_instnorm_act_fwd_t': (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _f, _f, _u64, _p, _sz, _p, _i]),
    'pg_instnorm_act_fwd_parts_t': (_i, [_p, _i, _p, _i, _p, _p, _i, _i, _i, _i, _i, _f, _f, _u64, _p, _i]),
    'pg_instnorm_act_bwd_t': (_i, [_p, _i, _p, _i, _p, _i, _p, _p, _i, _i, _i, _i, _i, _f, _u64, _p, _sz, _p, _i]),
    'pg_act_fwd_t': (_i, [_p, _i, _p, _i, _l, _i, _i, _f, _u64, _p, _i]),
    'pg_act_bwd_t': (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _l, _i, _i, _f, _u64, _p, _i]),
    'pg_softmax_fwd': (_i, [_p, _i, _p, _i, _l, _i, _p]),
    'pg_softmax_bwd': (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _l, _i, _p]),
    'pg_dropout_mask': (_i, [_p, _l, _f, _u64, _p]),
    'pg_loss_reduce_doubles': (_l, [_i, _i, _i]),
    'pg_loss_fused_max_nc': (_i, []),
    'pg_loss_reduce': (_i, [_p, _i, _p, _i, _f, _i, _i, _i, _p, _p]),
    'pg_loss_prepare': (_i, [_p, _i, _i, _f, _p, _p]),
    'pg_loss_finalize': (_i, [_p, _p, _i, _i, _i, _i, _i, _f, _f, _f, _p, _p, _p]),
    'pg_loss_grad': (_i, [_p, _i, _p, _i, _f, _p, _p, _i, _i, _i, _i, _i, _p]),
    'pg_loss_reduce_parts': (_i, [_p, _i, _p, _i, _f, _i, _i, _i, _p, _p]),
    'pg_loss_value_grad': (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _f, _f, _f, _p, _i, _p, _i, _f, _p, _i, _p, _p]),
    'pg_adam_step': (_i, [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _f, _p]),
    'pg_adam_step_dev': (_i, [_p, _p, _p, _p, _l, _f, _f, _f, _p, _p]),
    'pg_nchw_to_nhwc': (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    'pg_nhwc_to_nchw': (_i, [_p, _i, _p, _i, _i, _i, _i, _p]),
    'pg_copy_channels': (_i, [_p, _i, _p, _i, _l, _i, _p]),
    'pg_fill': (_i, [_p, _l, _f, _p]),
    'pg_pad8_bf16': (_i, [_p, _i, _p, _l, _i, _p]),
    'pg_din_fill': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    'pg_u8_to_f32': (_i, [_p, _p, _i, _l, _i, _f, _p]),
    'pg_labels_to_onehot': (_i, [_p, _p, _i, _l, _p, _i, _i, _p]),
    'pg_tiles_count': (_i, [_i, _i, _i]),
    'pg_tiles_gather': (_i, [_p, _i, _i, _i, _i, _i, _p, _i, _p]),
    'pg_tiles_blend': (_i, [_p, _i, _i, _i, _i, _i, _i, ctypes.c_double, _p, _p, _p]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """Load the shared library once; raise HipLibraryError (never fall back) if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C patchgan_amd/csrc` (hipcc --offload-arch=gfx950). patchgan_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {_ERR.get(rc, rc)}")


def ptr(t, offset=0):
    """Device address of element `offset` (in elements) of a torch tensor, or None."""
    if t is None:
        return None
    return t.data_ptr() + offset * t.element_size()
