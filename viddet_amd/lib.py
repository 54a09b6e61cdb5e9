"""ctypes binding of libviddet_hip.so (C-ABI declared in include/viddet_hip.h).

The product path has no CPU fallback: if the shared library is missing or a call fails, this
module raises.  Tensors are torch tensors used as storage only (``data_ptr()`` + current stream).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VD_LIB: developer override for A/B-testing a differently built kernel library (tools/ only)
LIB_PATH = os.environ.get("VD_LIB") or os.path.join(_HERE, "csrc", "libviddet_hip.so")

ABI_VERSION = 6          # include/viddet_hip.h VD_ABI_VERSION
VD_MAX_TAPS = 27
EPI_AFFINE, EPI_LEAKY, EPI_RESIDUAL = 1, 2, 4
MATH_SPLIT = 16        # vd_conv_desc.flags / vd_wgrad_desc.flags: split-operand fp32 products (include/viddet_hip.h)
MATH_BF16 = 32         # products on bf16-rounded operands (one MFMA term), fp32 tensors and accumulation
MATH_F16X2 = 64        # two-way fp16 operand split with per-tensor power-of-two scales (three MFMA terms, fp32-accurate)
MATH_NOHALO = 128      # with MATH_F16X2: generic K loop instead of the halo-staged one (A/B timing)
STORE_BF16 = 256       # vd_wgrad_desc.flags: `in` / `dout` are bf16 tensors (bf16-storage training)
WGRAD_HALO = 512       # vd_wgrad_desc.flags: halo-ring kernel for 3x3 / stride-1 weight gradients (vd_wgrad_halo.hip)
CONV_STREAMK = 1024    # vd_conv_desc.flags: persistent stream-K grid (vd_conv_sk.hip); bit-identical results
CONV_SPLITK = 4096     # vd_conv_desc.flags (vd_conv_igemm_bf16): split-K for launches with too few tiles; deterministic, not bit-identical
CONV_PARITY4 = 2048    # vd_conv_desc.flags: a 3x3 / stride-2 data gradient as ONE launch (vd_conv_par.hip)
SK_HEADER_BYTES = 32768
AMAX_SLOTS, AMAX_STRIDE = 32, 64
AMAX_FLOATS = AMAX_SLOTS * AMAX_STRIDE      # floats of one tensor's max-abs slots (include/viddet_hip.h)

_fp = C.c_void_p


class ConvDesc(C.Structure):
    _fields_ = [
        ("in_", _fp), ("wp", _fp), ("out", _fp), ("scale", _fp), ("shift", _fp), ("residual", _fp),
        ("N", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32), ("Ci", C.c_int32),
        ("Hg", C.c_int32), ("Wg", C.c_int32), ("in_stride", C.c_int32), ("T", C.c_int32),
        ("dy", C.c_int32 * VD_MAX_TAPS), ("dx", C.c_int32 * VD_MAX_TAPS), ("dz", C.c_int32 * VD_MAX_TAPS),
        ("Kfr", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32), ("Co", C.c_int32),
        ("out_stride", C.c_int32), ("out_oy", C.c_int32), ("out_ox", C.c_int32),
        ("ldo", C.c_int32), ("ldr", C.c_int32), ("flags", C.c_int32), ("slope", C.c_float),
        ("in_scale", _fp), ("in_shift", _fp), ("in_slope", C.c_float), ("tile", C.c_int32),
        ("stats_part", _fp),
        ("bs_z", _fp), ("bs_scale", _fp), ("bs_shift", _fp), ("bs_mean", _fp), ("bs_invstd", _fp), ("bs_part", _fp),
        ("bs_slope", C.c_float),
        ("amax_in", _fp), ("amax_w", _fp), ("amax_out", _fp),
        ("sk_ws", _fp), ("sk_ws_bytes", C.c_int64),
        ("par_cin", C.c_int32), ("par_mask", C.c_int32),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("in_", _fp), ("dout", _fp), ("dwp", _fp),
        ("N", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32), ("Ci", C.c_int32),
        ("Hg", C.c_int32), ("Wg", C.c_int32), ("Co", C.c_int32), ("ldd", C.c_int32),
        ("in_stride", C.c_int32), ("T", C.c_int32),
        ("dy", C.c_int32 * VD_MAX_TAPS), ("dx", C.c_int32 * VD_MAX_TAPS), ("dz", C.c_int32 * VD_MAX_TAPS),
        ("Kfr", C.c_int32), ("splits", C.c_int32),
        ("in_scale", _fp), ("in_shift", _fp), ("in_slope", C.c_float), ("flags", C.c_int32),
        ("amax_in", _fp), ("amax_dout", _fp),
    ]


class GuardItem(C.Structure):
    _fields_ = [("gamma", _fp), ("beta", _fp), ("scale", _fp), ("C", C.c_int32), ("pad_", C.c_int32)]


class HeadDesc(C.Structure):
    _fields_ = [
        ("head", _fp * 3), ("g", C.c_int32 * 3), ("ldh", C.c_int32),
        ("stride", C.c_float * 3), ("anchors", (C.c_float * 6) * 3),
        ("B", C.c_int32), ("C", C.c_int32),
    ]


# every symbol include/viddet_hip.h declares, with its ctypes signature (restype, argtypes)
_i, _i64, _f, _d, _p = C.c_int, C.c_int64, C.c_float, C.c_double, C.c_void_p
SIGNATURES = {
    "vd_last_error": (C.c_char_p, []),
    "vd_version": (_i, []),
    "vd_abi_version": (_i, []),
    "vd_sizeof_desc": (_i64, [_i]),
    "vd_conv_igemm": (_i, [C.POINTER(ConvDesc), _p]),
    "vd_conv_igemm_mtiles": (_i, [C.POINTER(ConvDesc)]),
    "vd_conv_igemm_streamk": (_i, [C.POINTER(ConvDesc)]),
    "vd_conv_igemm_streamk_ws_bytes": (_i64, []),
    "vd_conv_igemm_bf16": (_i, [C.POINTER(ConvDesc), _i, _p]),
    "vd_conv_igemm_bf16_mtiles": (_i, [C.POINTER(ConvDesc)]),
    "vd_conv_igemm_bf16_streamk": (_i, [C.POINTER(ConvDesc), _i]),
    "vd_pack_weight_bf16": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "vd_stem_im2col_bf16": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "vd_conv_wgrad_ws_bytes": (_i64, [C.POINTER(WgradDesc)]),
    "vd_conv_wgrad": (_i, [C.POINTER(WgradDesc), _p, _i64, _p]),
    "vd_conv_wgrad_uses_halo": (_i, [C.POINTER(WgradDesc)]),
    "vd_stem_im2col": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "vd_pack_weight_fwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "vd_pack_weight_dgrad": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_int32), _i, _i, _p]),
    "vd_pack_weight_dgrad_s2": (_i, [_p, _p, _i, _i, _i, C.POINTER(C.c_int32), _p]),
    "vd_unpack_wgrad": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "vd_bn_stats_ws_bytes": (_i64, [_i64, _i]),
    "vd_bn_stats": (_i, [_p, _i64, _i, _p, _p, _i64, _p]),
    "vd_bn_sum_partials_ws_bytes": (_i64, [_i, _i]),
    "vd_bn_sum_partials": (_i, [_p, _i, _i, _p, _p, _i64, _p]),
    "vd_bn_finalize": (_i, [_p, _d, _i, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _p]),
    "vd_bn_sum_finalize": (_i, [_p, _i, _i, _p, _d, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _p, _i64, _p]),
    "vd_bn_sum_param_grads": (_i, [_p, _i, _i, _p, _p, _p, _p, _i64, _p]),
    "vd_bn_fold_eval": (_i, [_p, _p, _p, _p, _f, _i, _p, _p, _p]),
    "vd_bn_apply_leaky": (_i, [_p, _p, _p, _p, _p, _i64, _i, _f, _p, _p]),
    "vd_bn_bwd_reduce": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i, _f, _p, _p, _i64, _p]),
    "vd_bn_param_grads": (_i, [_p, _i, _p, _p, _p]),
    "vd_bn_bwd_apply": (_i, [_p, _p, _p, _p, _p, _p, _p, _d, _i64, _i, _f, _p, _p, _p]),
    "vd_range_guard": (_i, [_p, _i, _f, _p, _p, _p]),
    "vd_amax": (_i, [_p, _i64, _p, _p]),
    "vd_amax_segments": (_i, [_p, _p, _i, _p, _p]),
    "vd_amax_merge": (_i, [_p, _p, _p, _p]),
    "vd_add": (_i, [_p, _p, _p, _i64, _p]),
    "vd_fill": (_i, [_p, _f, _i64, _p]),
    "vd_upsample2x_concat": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vd_upsample2x_concat_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vd_nchw_to_nhwc": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "vd_stem_conv_blocks": (_i, [_i, _i, _i]),
    "vd_stem_conv": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p, C.c_float, _i, _i, _p, _p]),
    "vd_stem_conv_c32_bf16": (_i, [_p, _p, _p, _p, C.c_float, C.POINTER(ConvDesc), _p]),
    "vd_stem_wgrad_ws_bytes": (_i64, [_i, _i, _i]),
    "vd_stem_wgrad": (_i, [_p, _p, _i, _p, _i, _i, _i, _p, _i64, _p]),
    "vd_preprocess_u8_nhwc": (_i, [_p, _p, _i64, _p]),
    "vd_preprocess_u8_nchw": (_i, [_p, _p, _i, _i, _i, _p]),
    "vd_temporal_pool": (_i, [_p, _p, _p, _i, _i, _i64, _i, _p]),
    "vd_temporal_pool_bwd": (_i, [_p, _p, _p, _i, _i, _i64, _i, _p]),
    "vd_temporal_cat": (_i, [_p, _p, _i, _i, _i64, _i, _i, _p]),
    "vd_frame_slice": (_i, [_p, _p, _i, _i, _i, _i, _i64, _i, _p]),
    "vd_yolo_decode_filter": (_i, [C.POINTER(HeadDesc), _f, _p, _p, C.c_int32, _p, _p]),
    "vd_nms_ws_bytes": (_i64, [_i, _i, _i]),
    "vd_nms_topk": (_i, [C.POINTER(HeadDesc), _p, _p, C.c_int32, _p, _f, _i, _i, _p, _p, _p, _p, _p, _i64, _p]),
    "vd_yolo_loss_ws_bytes": (_i64, [C.POINTER(HeadDesc)]),
    "vd_yolo_loss_fwd_bwd": (_i, [C.POINTER(HeadDesc), _p, _i, _p, _p, _p, _p, _p, _f, _i, _p,
                                  C.POINTER(_fp * 3), _p, C.POINTER(_fp * 3), _p, _i64, _p]),
    "vd_sgd_momentum": (_i, [_p, _p, _p, _i64, _f, _f, _f, _f, _p]),
    # bf16-storage training (include/viddet_hip.h, last section)
    "vd_bn_stats_bf16": (_i, [_p, _i64, _i, _p, _p, _i64, _p]),
    "vd_bn_apply_leaky_bf16": (_i, [_p, _p, _p, _p, _p, _i64, _i, _f, _p]),
    "vd_bn_bwd_reduce_bf16": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i, _f, _p, _p, _i64, _p]),
    "vd_bn_bwd_apply_bf16": (_i, [_p, _p, _p, _p, _p, _p, _p, _d, _i64, _i, _f, _p, _p]),
    "vd_add_bf16": (_i, [_p, _p, _p, _i64, _p]),
    "vd_temporal_pool_bf16": (_i, [_p, _p, _i, _i, _i64, _i, _p]),
    "vd_pack_weight_dgrad_bf16": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_int32), _i, _i, _p]),
    "vd_upsample2x_concat_bwd_bf16": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vd_stem_wgrad_bf16": (_i, [_p, _p, _i, _p, _i, _i, _i, _p, _i64, _p]),
    "vd_yolo_loss_fwd_bwd_bf16": (_i, [C.POINTER(HeadDesc), _p, _i, _p, _p, _p, _p, _p, _f, _i, _p,
                                       C.POINTER(_fp * 3), _p, _p, _i64, _p]),
}

_lib = None


class VidDetHipError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VidDetHipError(
            "libviddet_hip.so not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C viddet_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    # torch bundles its own HIP runtime (libamdhip64.so): it must be resident BEFORE this library is
    # dlopen'ed so that both resolve to ONE runtime (the device memory torch allocates and the stream it
    # hands over must belong to the runtime that launches these kernels).  Loading /opt/rocm's copy first
    # yields a second runtime that reports "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    # the ABI revision and the descriptor layouts this binding was written against (include/viddet_hip.h VD_ABI_VERSION):
    # a library built from another header revision would misread positional pointers - refuse before the first launch
    if lib.vd_abi_version() != ABI_VERSION:
        raise VidDetHipError("%s has ABI revision %d, this binding is written for %d: rebuild (make -C viddet_amd/csrc)"
                             % (LIB_PATH, lib.vd_abi_version(), ABI_VERSION))
    for i, st in enumerate((ConvDesc, WgradDesc, HeadDesc)):
        if lib.vd_sizeof_desc(i) != C.sizeof(st):
            raise VidDetHipError("%s: sizeof(%s) is %d in the library, %d in this binding" %
                                 (LIB_PATH, st.__name__, lib.vd_sizeof_desc(i), C.sizeof(st)))
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        raise VidDetHipError("%s failed (rc=%d): %s" % (what, rc, load().vd_last_error().decode()))


def stream_ptr():
    """hipStream_t of torch's current stream, as an integer (kernels are ordered on it)."""
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())
