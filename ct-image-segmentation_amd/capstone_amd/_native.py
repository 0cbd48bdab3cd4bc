"""ctypes binding of libctseg_hip.so (the C ABI declared in include/ctseg_hip.h).

There is NO fallback: if the shared library is missing or a call is rejected this raises.
PyTorch is used only for device memory and streams; every pointer handed over is a raw
``data_ptr()`` and the stream is ``torch.cuda.current_stream().cuda_stream``.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# CTSEG_LIB: another build of the same library (tests/test_host_sanitizer.py loads the host-sanitizer build on the CPU); default and
# product path: the in-tree build
LIB_PATH = os.environ.get("CTSEG_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libctseg_hip.so")

F32, BF16, I16, U8, F16 = 0, 1, 2, 3, 4      # F16: IEEE half storage, forward (inference) passes only
MAX_TAPS, MAX_CLASSES = 27, 8
ABI_VERSION = 3              # CTSEG_ABI_VERSION of include/ctseg_hip.h this binding mirrors
_TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}
_EPC = {F32: 4, BF16: 8, F16: 8}
_SZ = {F32: 4, BF16: 2, F16: 2}
DT_OF_PRECISION = {"fp32": F32, "bf16": BF16, "fp16": F16}


def is16(dt):
    """16-bit storage kinds: same layouts (8 elements per 16-byte chunk, 12-wide rows around 10 classes), same kernels"""
    return dt in (BF16, F16)


def torch_dtype(dt):
    return _TORCH_DT[dt]


def epc(dt):
    """elements per 16-byte chunk"""
    return _EPC[dt]


def elsize(dt):
    return _SZ[dt]


class NativeError(RuntimeError):
    pass


class ConvClass(C.Structure):
    _fields_ = [("ntaps", C.c_int32), ("kpad", C.c_int32), ("w_off", C.c_int64),
                ("ox", C.c_int32), ("oy", C.c_int32), ("oz", C.c_int32), ("taps", C.c_int32 * MAX_TAPS)]


class _SizedDesc(C.Structure):
    """ABI 2: a descriptor starts with the size of the struct its caller was compiled against (checked by every entry point)"""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.struct_size = C.sizeof(self)


class ConvDesc(_SizedDesc):
    _fields_ = [("struct_size", C.c_int32), ("reserved0", C.c_int32), ("in_", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p),
                ("add", C.c_void_p), ("stats", C.c_void_p), ("dtype", C.c_int32),
                ("N", C.c_int32), ("Xi", C.c_int32), ("Yi", C.c_int32), ("Zi", C.c_int32),
                ("Xr", C.c_int32), ("Yr", C.c_int32), ("Zr", C.c_int32),
                ("Xo", C.c_int32), ("Yo", C.c_int32), ("Zo", C.c_int32),
                ("Cg", C.c_int32), ("Cn", C.c_int32), ("Cn_store", C.c_int32),
                ("g_ld", C.c_int32), ("o_ld", C.c_int32), ("add_ld", C.c_int32),
                ("sin", C.c_int32), ("sout", C.c_int32), ("out_f32", C.c_int32), ("add_f32", C.c_int32),
                ("stats_ld", C.c_int32), ("stats_tiles", C.c_int32), ("stats_tile0", C.c_int32),
                ("nclass", C.c_int32), ("cls", ConvClass * MAX_CLASSES),
                ("out2", C.c_void_p), ("out2_col0", C.c_int32), ("o2_ld", C.c_int32),
                ("in_mean_rstd", C.c_void_p), ("in_alpha", C.c_void_p), ("in_norm_C", C.c_int32),
                ("bst_y", C.c_void_p), ("bst_mean_rstd", C.c_void_p), ("bst_alpha", C.c_void_p), ("bst_partials", C.c_void_p),
                ("bst_y_ld", C.c_int32), ("bst_C", C.c_int32), ("bst_col0", C.c_int32), ("bst_P", C.c_int32), ("bst_ld", C.c_int32),
                ("reserved1", C.c_int32)]


class PackPart(C.Structure):
    _fields_ = [("o", C.c_int64), ("n_lo", C.c_int32), ("n_hi", C.c_int32), ("g_lo", C.c_int32), ("g_hi", C.c_int32),
                ("SN", C.c_int32), ("SG", C.c_int32)]


class PackBlock(C.Structure):
    _fields_ = [("dst_off", C.c_int64), ("kpad", C.c_int32), ("gs", C.c_int32), ("ntaps", C.c_int32), ("T", C.c_int32),
                ("nparts", C.c_int32), ("reserved", C.c_int32), ("part", PackPart * 2), ("tap", C.c_int32 * MAX_TAPS),
                ("reserved2", C.c_int32)]


PACK_LDS_FLOATS = 12288


class ReduceJob(C.Structure):
    """ctseg_reduce_job: one ctseg_conv_wgrad_reduce call of a batched launch"""
    _fields_ = [("ws", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p), ("nslabs", C.c_int32), ("kpad_w", C.c_int32),
                ("cn_pad", C.c_int32), ("A", C.c_int32), ("Astride", C.c_int32), ("T", C.c_int32), ("col0", C.c_int32),
                ("nb", C.c_int32), ("block0", C.c_int32), ("lanes", C.c_int32)]


class WgradDesc(_SizedDesc):
    _fields_ = [("struct_size", C.c_int32), ("reserved0", C.c_int32), ("in_", C.c_void_p), ("dy", C.c_void_p), ("ws", C.c_void_p), ("dtype", C.c_int32),
                ("N", C.c_int32), ("Xi", C.c_int32), ("Yi", C.c_int32), ("Zi", C.c_int32),
                ("Xr", C.c_int32), ("Yr", C.c_int32), ("Zr", C.c_int32),
                ("Cg", C.c_int32), ("Cn", C.c_int32), ("g_ld", C.c_int32), ("d_ld", C.c_int32), ("sin", C.c_int32),
                ("ntaps", C.c_int32), ("taps", C.c_int32 * MAX_TAPS), ("splits", C.c_int32),
                ("kpad_w", C.c_int32), ("cn_pad", C.c_int32),
                ("in_mean_rstd", C.c_void_p), ("in_alpha", C.c_void_p), ("in_norm_C", C.c_int32),
                ("dyn_col0", C.c_int32), ("dyn_g", C.c_void_p), ("dyn_y", C.c_void_p), ("dyn_g_ld", C.c_int32), ("dyn_y_ld", C.c_int32),
                ("dyn_mean_rstd", C.c_void_p), ("dyn_alpha", C.c_void_p), ("dyn_sums", C.c_void_p)]


_i32, _i64, _f32, _f64, _vp = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_void_p
_SIGS = {
    "ctseg_abi_version": (C.c_int, []),
    "ctseg_last_error": (C.c_char_p, []),
    "ctseg_conv_tile_rows": (C.c_int, [_i32]),
    "ctseg_conv_tile_cols": (C.c_int, [_i32]),
    "ctseg_conv_num_tiles": (C.c_int, [C.POINTER(ConvDesc)]),
    "ctseg_conv_split_ok": (C.c_int, [C.POINTER(ConvDesc)]),
    "ctseg_conv_narrow_ok": (C.c_int, [C.POINTER(ConvDesc)]),
    "ctseg_wgrad_narrow_ok": (C.c_int, [C.POINTER(WgradDesc)]),
    "ctseg_conv_in_norm_ok": (C.c_int, [C.POINTER(ConvDesc)]),
    "ctseg_conv_bwd_stats_slots": (C.c_int, [C.POINTER(ConvDesc)]),
    "ctseg_wgrad_in_norm_ok": (C.c_int, [C.POINTER(WgradDesc)]),
    "ctseg_wgrad_dy_norm_ok": (C.c_int, [C.POINTER(WgradDesc)]),
    "ctseg_conv_igemm": (C.c_int, [C.POINTER(ConvDesc), _vp]),
    "ctseg_wgrad_tile_cols": (C.c_int, [_i32]),
    "ctseg_conv_wgrad_slabs": (C.c_int, [C.POINTER(WgradDesc)]),
    "ctseg_conv_wgrad_wgs_per_slab": (C.c_int, [C.POINTER(WgradDesc), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ctseg_conv_wgrad": (C.c_int, [C.POINTER(WgradDesc), _vp]),
    "ctseg_conv_wgrad_reduce": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ctseg_conv_wgrad_reduce_batch_ok": (C.c_int, [_vp, _i32, _i32, _i32]),
    "ctseg_conv_wgrad_reduce_batch": (C.c_int, [_vp, _i32, _i32, _vp]),
    "ctseg_gather_cast": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _vp]),
    "ctseg_pack_weights": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _vp, _i32, _vp]),
    "ctseg_instnorm_finalize": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _f64, _f64, _vp, _vp, _vp]),
    "ctseg_instnorm_prelu_fwd": (C.c_int, [_i32, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _i64, _i32, _vp]),
    "ctseg_instnorm_prelu_bwd_reduce": (C.c_int, [_i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _i32, _vp]),
    "ctseg_instnorm_prelu_bwd_finalize": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _f64, _vp, _vp, _vp, _vp]),
    "ctseg_instnorm_prelu_dalpha": (C.c_int, [_vp, _i32, _vp, _vp]),
    "ctseg_instnorm_prelu_bwd_apply": (C.c_int, [_i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _i64, _i32,
                                                 _vp, _i32, _vp, _vp]),
    "ctseg_instnorm_prelu_bwd_apply_colsum": (C.c_int, [_i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _i64, _i32,
                                                        _vp, _i32, _vp, _vp, _i32, _vp, _vp]),
    "ctseg_colsum": (C.c_int, [_i32, _vp, _i32, _i64, _i32, _vp, _i32, _vp, _vp]),
    "ctseg_squash_masks": (C.c_int, [_vp, _i32, _i32, _i64, _vp, _vp, _vp, _vp]),
    "ctseg_seg_loss": (C.c_int, [_vp, _i32, _vp, _i32, _i64, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp]),
    "ctseg_conv_logits_ce_slots": (C.c_int, [C.POINTER(ConvDesc), _i32]),
    "ctseg_conv_logits_ce": (C.c_int, [C.POINTER(ConvDesc), _vp, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _vp, _vp]),
    "ctseg_dice_counts": (C.c_int, [_vp, _vp, _i32, _i64, _i32, _vp, _vp]),
    "ctseg_reduce_partials_f64": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "ctseg_loss_dice_summary": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _vp, _vp]),
    "ctseg_adam_step": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f64, _f64, _f64, _f64, _i32, _f32, _vp]),
    "ctseg_scale_inplace": (C.c_int, [_vp, _i32, _i64, _vp, _f32, _vp]),
    "ctseg_cast": (C.c_int, [_vp, _i32, _vp, _i32, _i64, _vp]),
    "ctseg_nc_to_cl": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i64, _i32, _vp]),
    "ctseg_cl_to_nc": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i64, _i32, _vp]),
    "ctseg_resize3d_to_hwd": (C.c_int, [_vp, _i32, _vp] + [_i32] * 8 + [_f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "ctseg_window_gather": (C.c_int, [_vp] + [_i32] * 10 + [_f32, _vp, _i32, _i32, _vp]),
    "ctseg_window_gather_batch": (C.c_int, [_vp] + [_i32] * 4 + [_vp] + [_i32] * 4 + [_f32, _vp, _i32, _i32, _vp]),
    "ctseg_window_blend": (C.c_int, [_vp] + [_i32] * 8 + [_vp, _vp, _vp] + [_i32] * 4 + [_vp]),
    "ctseg_window_blend_batch": (C.c_int, [_vp] + [_i32] * 5 + [_vp, _i32, _vp, _vp, _vp] + [_i32] * 4 + [_vp, _vp]),
}
EXPORTS = tuple(_SIGS)
_lib = None


def lib():
    """Load the shared library once; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(or `make -C ct-image-segmentation_amd`). There is no CPU/eager fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if L.ctseg_abi_version() != ABI_VERSION:
            raise NativeError(f"libctseg_hip.so reports ABI version {L.ctseg_abi_version()}, this binding is written against "
                              f"{ABI_VERSION} (include/ctseg_hip.h): rebuild with `make -C ct-image-segmentation_amd`")
        _lib = L
    return _lib


def query(name, desc):
    """host-only capability / sizing queries of the C ABI (tests route some of them through the ABI emulator)"""
    return getattr(lib(), name)(desc)


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def check(rc, what):
    if rc != 0:
        raise NativeError(f"{what} failed ({rc}): {lib().ctseg_last_error().decode()}")


def call(name, *args):
    check(getattr(lib(), name)(*args, stream_ptr()), name)


def ptr(t, byte_offset=0):
    return None if t is None else t.data_ptr() + byte_offset


def require_gpu(t, what):
    if not t.is_cuda:
        raise NativeError(f"{what}: tensor is on {t.device}; the MI355X path needs a ROCm device tensor "
                          "(there is no CPU fallback in the product path)")
