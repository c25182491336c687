"""ctypes binding of libtedspad_hip.so (C ABI: include/tedspad_hip.h).

The HIP library is the ONLY compute path of this package: if it is missing, loading fails
loudly -- there is no eager/PyTorch/CPU fallback (a fallback would void every parity claim).
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libtedspad_hip.so")

F16, BF16, F32 = 0, 1, 2
ABI_VERSION = 5          # TEDSPAD_ABI_VERSION of include/tedspad_hip.h this binding was written against


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n", "t", "h", "w", "cin", "ldx", "cout", "ldy", "ldres",
        "kt", "kh", "kw", "st", "sh", "sw", "pt", "ph", "pw",
        "to", "ho", "wo", "relu", "dtype", "tile_cfg")]


class ConvExtras(C.Structure):
    _fields_ = [("mask", C.c_void_p), ("stats", C.c_void_p), ("y32", C.c_void_p)] + [(n, C.c_int32) for n in (
        "ldmask", "stats_ld", "ldy32", "out_strided", "ost", "osh", "osw", "oot", "ooh", "oow", "tf", "hf", "wf", "fold_hw", "fold_c", "fold_ldy", "stats_rows", "nosat", "nchunk_src", "chunk_up")] + [
        ("chunk_ld", C.c_int32 * 8), ("chunk_src", C.c_void_p * 8)]


class PackJob(C.Structure):              # tedspad_pack_job
    _fields_ = [("w", C.c_void_p), ("scale", C.c_void_p), ("out", C.c_void_p)] + [(n, C.c_int32) for n in (
        "co", "ci", "kt", "kh", "kw", "cink", "kwk", "pair_shift", "mode", "rows", "rows_pad", "kpad")] + [
        ("geo", C.c_int32 * 9), ("dtype", C.c_int32), ("block0", C.c_int32), ("nblocks", C.c_int32)]


class FoldJob(C.Structure):              # tedspad_fold_job
    _fields_ = [(n, C.c_void_p) for n in ("gamma", "beta", "mean", "var", "conv_bias", "scale", "shift", "scale2", "shift2")] + [
        ("eps", C.c_double), ("C", C.c_int32), ("n", C.c_int32), ("n2", C.c_int32), ("reserved", C.c_int32)]


class WgradUnpackJob(C.Structure):       # tedspad_wgrad_unpack_job
    _fields_ = [("dw", C.c_void_p), ("grad", C.c_void_p), ("row_scale", C.c_void_p)] + [(n, C.c_int32) for n in (
        "co", "ci", "kt", "kh", "kw", "cink", "kpad", "accumulate", "block0", "nblocks")]


class PoolDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n", "t", "h", "w", "c", "ldx", "ldy",
        "kt", "kh", "kw", "st", "sh", "sw", "pt", "ph", "pw",
        "to", "ho", "wo", "pad_zero", "dtype")]


# name -> (restype, argtypes); must list every symbol include/tedspad_hip.h declares
_P, _I32, _I64 = C.c_void_p, C.c_int32, C.c_int64
SYMBOLS = {
    "tedspad_abi_version": (_I32, []),
    "tedspad_last_error": (C.c_char_p, []),
    "tedspad_conv_kpad": (_I32, [C.POINTER(ConvDesc)]),
    "tedspad_conv_cout_pad": (_I32, [C.POINTER(ConvDesc)]),
    "tedspad_conv_num_tile_cfgs": (_I32, []),
    "tedspad_conv_ktab_entries": (_I32, [C.POINTER(ConvDesc)]),
    "tedspad_conv_build_ktab": (_I32, [C.POINTER(ConvDesc), _P]),
    "tedspad_conv_fwd": (_I32, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _I32, _P]),
    "tedspad_conv_fwd_ex": (_I32, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _I32, C.POINTER(ConvExtras), _P]),
    "tedspad_conv_wgrad": (_I32, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P]),
    "tedspad_maxpool_fwd": (_I32, [C.POINTER(PoolDesc), _P, _P, _P]),
    "tedspad_maxpool_fwd_idx": (_I32, [C.POINTER(PoolDesc), _P, _P, _P, _P]),
    "tedspad_global_avgpool_fwd": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "tedspad_avgpool3d_s1_fwd": (_I32, [_P, _P] + [_I32] * 10 + [_P]),
    "tedspad_clip_to_channels_last": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _I32, _I64, _I64, _I64, _I64, _I64, _I32, _I32, _P]),
    "tedspad_channels_last_to_nchw": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "tedspad_upsample_bilinear2x_fwd": (_I32, [_P, _P] + [_I32] * 11 + [_P]),
    "tedspad_ntxent_fwd_bwd": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, C.c_float, _I32, _P]),
    "tedspad_triplet_fwd_bwd": (_I32, [_P] * 8 + [_I32, _I32, C.c_float, C.c_float, _P]),
    "tedspad_cross_entropy_fwd_bwd": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _P]),
    "tedspad_bn_finalize": (_I32, [_P, _I32, _I64, _P, _P, C.c_float, C.c_float, _P, _P, _P, _P, _P, _P, _I32, _P]),
    "tedspad_bn_train_apply": (_I32, [_P, _I32, _P, _I32, _I64, _P, _P, C.c_float, C.c_float, _P, _P, _P, _P, _I32, _P, _P, _I64] + [_I32] * 7 + [_P]),
    "tedspad_scale_shift_act": (_I32, [_P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "tedspad_bn_bwd_reduce": (_I32, [_P, _P, _P, _I32, _P, _P, _P, _P, _P, _I32, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "tedspad_bn_bwd_apply": (_I32, [_P, _P, _P, _I32] + [_P] * 5 + [_I32, _P, _P, _P, _I32, _I64] + [_I32] * 9 + [_P]),
    "tedspad_maxpool_bwd": (_I32, [C.POINTER(PoolDesc), _P, _P, _P, _I32, _P, _I32, _P, _I32, _I32, _P]),
    "tedspad_global_avgpool_bwd": (_I32, [_P, _P, _I32, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "tedspad_upsample_bilinear2x_bwd": (_I32, [_P, _P] + [_I32] * 11 + [_P]),
    "tedspad_nchw_grad_to_channels_last": (_I32, [_P, _P, _P, _I32, _I32, _I64, _I32, _P]),
    "tedspad_channels_last_to_nchw_strided": (_I32, [_P, _P] + [_I32] * 6 + [_I64] * 5 + [_I32, _P]),
    "tedspad_bn1d_train_fwd": (_I32, [_P, _P, _P, C.c_float, C.c_float, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P]),
    "tedspad_bn1d_train_bwd": (_I32, [_P] * 9 + [_I32, _I32, _I32, _P]),
    "tedspad_l2_normalize_rows_bwd": (_I32, [_P, _P, _P, _I32, _I32, C.c_float, _P]),
    "tedspad_mul_f32": (_I32, [_P, _P, _P, _I64, C.c_float, _P]),
    "tedspad_pack_conv_weights": (_I32, [_P, _P, _P] + [_I32] * 12 + [_P, _I32, _P]),
    "tedspad_linear_fwd": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "tedspad_l2_normalize_rows": (_I32, [_P, _P, _I32, _I32, C.c_float, _P]),
    "tedspad_conv_pool_t2_fwd": (_I32, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P]),
    "tedspad_conv_pw_dual_fwd": (_I32, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _I32, _P, _P, _P, _P, _P]),
    "tedspad_conv_p8_dual_fwd": (_I32, [C.POINTER(ConvDesc), _P, _P] + [_I32] * 6 + [_P, _P, _P, _P, _P]),
    "tedspad_bneck_frame_fwd": (_I32, [_P, _I32, _P, _I32] + [_I32] * 6 + [_P, _P, _I32, _P] + [_P] * 6 + [_I32, _I32, _P]),
    "tedspad_bneck_frame_lds_bytes": (_I32, []),
    "tedspad_clock_probe": (_I32, [_I32, _I32, _P, _P]),
    "tedspad_set_deterministic": (_I32, [_I32]),
    "tedspad_deterministic_giveups": (_I32, []),
    "tedspad_bneck_tail_fwd": (_I32, [C.POINTER(ConvDesc)] + [_P] * 7 + [_I32, _P, _I32, _P, _I32, _P, _P, _I32, _I32, _I32, _P]),
    "tedspad_clip_to_tp": (_I32, [_P, _P] + [_I32] * 5 + [_I64] * 5 + [_I32] * 4 + [_P]),
    "tedspad_stem_pt_wimg_bytes": (_I32, []),
    "tedspad_stem_pt_wimg16_bytes": (_I32, []),
    "tedspad_stem_pt_fwd": (_I32, [_P] * 5 + [_I32] * 11 + [_P]),
    "tedspad_stem_pt_side_bytes": (_I64, [_I32] * 4),
    "tedspad_stem_pt_pool_fwd": (_I32, [_P] * 6 + [_I32] * 10 + [_P]),
    "tedspad_unetpp_tail_wimg_bytes": (_I32, []),
    "tedspad_unetpp_tail_fwd": (_I32, [_P, _I32, _P, _I32, _I32, _I32] + [_P] * 6 + [_I32, _P]),
    "tedspad_stem_pt_pool_clip_fwd": (_I32, [_P] + [_I32] * 5 + [_I64] * 5 + [_I32] * 3 + [_P] * 5 + [_I32] * 6 + [_P]),
    "tedspad_upsample_nearest2x_fwd": (_I32, [_P, _P] + [_I32] * 6 + [_P]),
    "tedspad_copy_channels": (_I32, [_P, _P, _I64, _I32, _I32, _I32, _P]),
    "tedspad_upsample_nearest2x_bwd": (_I32, [_P, _P] + [_I32] * 8 + [_P]),
    "tedspad_add_channels": (_I32, [_P, _P, _I64, _I32, _I32, _I32, _I32, _P]),
    "tedspad_bn_fold": (_I32, [_P, _P, _P, _P, _P, C.c_double, _I32, _P, _P, _P]),
    "tedspad_pack_multi": (_I32, [_P, _I32, _P, _I32, _P]),
    "tedspad_fold_multi": (_I32, [_P, _I32, _P, _I32, _P]),
    "tedspad_wgrad_unpack_multi": (_I32, [_P, _I32, _P, _I32, _P]),
    "tedspad_resize_aa_taps": (_I32, [_I32, _I32]),
    "tedspad_resize_aa_table": (_I32, [_I32, _I32, _P]),
    "tedspad_frames_crop_resize": (_I32, [_P] + [_I32] * 11 + [_P, _P, C.c_float, _I32, _P] + [_I64] * 4 + [_P]),
    "tedspad_frames_crop_resize_tp": (_I32, [_P] + [_I32] * 16 + [_P, _P, C.c_float, _I32, _P] + [_I32] * 4 + [_P]),
    "tedspad_frames_crop_resize_pil": (_I32, [_P] + [_I32] * 10 + [_P, _I32, _P, _I32, _P] + [_I64] * 4 + [_P]),
    "tedspad_segment_pool_mag": (_I32, [_P, _I32, _I32, _I32, _I32, _P, _P]),
    "tedspad_count_saturated": (_I32, [_P, _I64, _I32, _I32, _I32, _P, _P]),
}

_lib = None


class TedSpadHipError(RuntimeError):
    pass


def lib():
    """The loaded library; raises if it has not been built (python -m ted_spad_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TedSpadHipError(
                "libtedspad_hip.so not found at %s -- the HIP extension is the only compute path "
                "(no CPU/PyTorch fallback). Build it: python -c 'import __graft_entry__ as g; g.build()'" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(l, name)  # AttributeError if the .so lacks a declared symbol
            f.restype = res
            f.argtypes = args
        if l.tedspad_abi_version() != ABI_VERSION:
            raise TedSpadHipError("libtedspad_hip.so ABI version mismatch")
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().tedspad_last_error()
        raise TedSpadHipError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))
