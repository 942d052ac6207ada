"""ctypes binding of libstar_hip.so (declared in include/star_hip.h).

The product path has no CPU fallback: if the shared library is missing or a call fails, an exception is
raised.  Tensors are passed as raw device pointers plus sizes; kernels are enqueued on torch's current
HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STAR_HIP_LIB") or os.path.join(_HERE, "libstar_hip.so")   # override: A/B builds of the kernels

BF16, F32, BF16X3 = 0, 1, 2     # BF16X3: bf16 hi/lo pairs, three MFMA products per multiply (forward only; include/star_hip.h)
ACT_NONE, ACT_GELU, ACT_RELU, ACT_DGELU, ACT_DRELU, ACT_MULAUX = 0, 1, 2, 3, 4, 5
GEMM_RES_BY_AROW = 1
GEMM_PRE_IS_GRAD = 2
GEMM_A_PLAIN = 4

_lib = None


class StarGemm(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("a_rows", C.c_void_p), ("c_rows", C.c_void_p),
        ("bias", C.c_void_p), ("res", C.c_void_p), ("res_mod", C.c_int32),
        ("res2", C.c_void_p), ("res2_div", C.c_int32), ("res2_mod", C.c_int32),
        ("aux", C.c_void_p), ("act", C.c_int32), ("pre", C.c_void_p), ("out_lp", C.c_void_p),
        ("out_f32", C.c_void_p), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("dt", C.c_int32),
        ("flags", C.c_int32), ("conv", C.c_int32),
    ]


class StarWgradJob(C.Structure):
    _fields_ = [("dY", C.c_void_p), ("X", C.c_void_p), ("dW", C.c_void_p), ("db", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("tap", C.c_int32)]


WGRAD_GROUP_MAX = 8
_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
_SIGS = {
    "star_version": (C.c_int, []),
    "star_last_error": (C.c_char_p, []),
    "star_warp_fuse": (C.c_int, [_P, _P, _L, _L, _L, _L, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "star_warp_fuse_ws": (C.c_int64, [_I, _I, _I, _I]),
    "star_mask_sort": (C.c_int, [_P, _P, _P, _I, _I, _P]),
    "star_build_rowmaps": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "star_layernorm_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _F, _I, _P]),
    "star_layernorm_bwd_ws": (C.c_int64, [_I]),
    "star_layernorm_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "star_layernorm_fold_rows": (C.c_int, [_I]),
    "star_layernorm_fold_batch": (C.c_int, [_P, _I, _I, _P]),
    "star_layernorm_bwd_static": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "star_gemm": (C.c_int, [C.POINTER(StarGemm), _P]),
    "star_gemm_wgrad_ws": (C.c_int64, [_I, _I, _I]),
    "star_gemm_wgrad": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "star_gemm_wgrad_group_ws": (C.c_int64, [C.POINTER(StarWgradJob), _I, _I]),
    "star_gemm_wgrad_group": (C.c_int, [C.POINTER(StarWgradJob), _I, _P, _L, _I, _P]),
    "star_attn_fwd": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "star_attn_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "star_gemm_res_ln_supported": (C.c_int, [_I, _I, _I]),
    "star_gemm_res_ln_prefers": (C.c_int, [_I, _I, _I, _I]),
    "star_gemm_res_ln": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_float, _P, _P, _P, _I, _I, _I, _I, _P]),
    "star_patchify": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "star_ce_nblocks": (C.c_int, [_I, _I, _I, _I]),
    "star_ce_argmax": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "star_ce_finish": (C.c_int, [_P, _I, _L, _P, _P]),
    "star_fill_rows": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "star_transpose_rows": (C.c_int, [_P, _P, _P, _I, _I, _I, _P]),
    "star_transpose_rows_bwd": (C.c_int, [_P, _P, _I, _I, _I, _P]),
    "star_gather_rows": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "star_scatter_rows": (C.c_int, [_P, _I, _P, _P, _I, _I, _P]),
    "star_colsum_groups": (C.c_int, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "star_cast_transpose": (C.c_int, [_P, _P, _I, _I, _I, _P]),
    "star_cast": (C.c_int, [_P, _P, _L, _I, _P]),
    "star_cast_back": (C.c_int, [_P, _P, _L, _I, _P]),
    "star_cast_batch": (C.c_int, [_P, _I, _I, _I, _P]),
    "star_adamw": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _P]),
    "star_sumsq": (C.c_int, [_P, _L, _P, _P, _P]),
    "star_iou_confusion": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "star_vq_nblocks": (C.c_int, [_I]),
    "star_vq_assign": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "star_vq_finish": (C.c_int, [_P, _I, _P, _I, _I, _I, _F, _I, _P, _P, _P]),
    "star_vq_code_sums": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "star_vq_ema_update": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, C.c_double, C.c_double, _P]),
    "star_vq_bwd_ws": (C.c_int64, [_I, _I, _I]),
    "star_vq_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _P]),
    "star_onehot_rows": (C.c_int, [_P, _P, _I, _I, _P]),
    "star_voxel_scatter": (C.c_int, [_P, _I, _P, _P, _L, _P, _I, _I, _I, _I, _I, _P, _P]),
    "star_feat_warp_fuse": (C.c_int, [_P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "star_upsample2x_fwd": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "star_upsample2x_bwd": (C.c_int, [_P, _P, _I, _I, _I, _I, _P]),
    "star_bn_ws": (C.c_int64, [_I]),
    "star_bn_stats": (C.c_int, [_P, _L, _I, _F, _P, _P, _P, _P, _P]),
    "star_bn_apply": (C.c_int, [_P, _P, _P, _P, _P, _I, _L, _I, _P, _I, _P]),
    "star_bn_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _L, _I, _P, _I, _P, _P, _P, _P]),
    "star_ce_nhwc_blocks": (C.c_int, [_I, _I, _I]),
    "star_ce_nhwc": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
}
EXPORTS = tuple(_SIGS)


def lib():
    """Load libstar_hip.so once.  Raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("star_amd: %s is missing -- build it with `make -C star_amd/csrc` "
                               "(there is no CPU fallback for the HIP path)" % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError("libstar_hip: " + lib().star_last_error().decode())


def stream():
    return torch.cuda.current_stream().cuda_stream


def p(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def strides(t):
    """int64 ctypes array of element strides."""
    return (C.c_int64 * t.dim())(*t.stride())


def dt_of(t):
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise TypeError("activation dtype must be bfloat16 or float32, got %s" % t.dtype)


def torch_dtype(dt):
    return torch.float32 if dt == F32 else torch.bfloat16


def width(dt):
    """stored elements per logical element of an activation / weight row (2 for the hi/lo pairs of BF16X3)"""
    return 2 if dt == BF16X3 else 1
