"""ctypes binding of libsr3hip.so (include/sr3hip.h). No torch types cross this boundary.

The product path has no CPU fallback: if the library is missing `load()` raises, and every op
raises `Sr3Error` with the library's message on a non-zero return code.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# SR3_LIB: another build of the same C-ABI (development: the -DSR3_EXPERIMENTS library of build.py --experiments)
LIB_PATH = os.environ.get("SR3_LIB") or os.path.join(HERE, "libsr3hip.so")

SR3_MAX_MULTS = 8
SR3_MAX_ATTN_RES = 8
N_FAMILIES = 5
FAMILIES = ("conv_igemm", "groupnorm", "attention", "embed", "update_layout")


class Sr3Error(RuntimeError):
    pass


class Sr3RangeWarning(RuntimeWarning):
    """A call in split-f16 mode left the fp16 range and was finished in exact f32 (SR3_OK_F32_FALLBACK)."""


class Sr3ReplayWarning(RuntimeWarning):
    """Part of a call was replayed on the conv path without inter-block waits because another kernel held the compute
    units an in-place split-K conv waits on (SR3_OK_REPLAYED); the result is complete and valid."""


class UnetCfg(C.Structure):
    _fields_ = [
        ("in_channel", C.c_int32),
        ("out_channel", C.c_int32),
        ("inner_channel", C.c_int32),
        ("norm_groups", C.c_int32),
        ("n_mults", C.c_int32),
        ("channel_mults", C.c_int32 * SR3_MAX_MULTS),
        ("n_attn_res", C.c_int32),
        ("attn_res", C.c_int32 * SR3_MAX_ATTN_RES),
        ("res_blocks", C.c_int32),
        ("image_size", C.c_int32),
        ("dropout", C.c_float),
    ]


_P = C.c_void_p
_F = C.c_void_p          # float* (device or host) passed as an integer address
_I = C.c_int
_U64 = C.c_uint64

# name -> (restype, argtypes); every symbol declared in include/sr3hip.h
PROTOTYPES = {
    "sr3_create": (_I, [C.POINTER(UnetCfg), _I, C.POINTER(_P)]),
    "sr3_destroy": (None, [_P]),
    "sr3_last_error": (C.c_char_p, []),
    "sr3_set_stream": (_I, [_P, _P]),
    "sr3_synchronize": (_I, [_P]),
    "sr3_wait_for_stream": (_I, [_P, _P]),
    "sr3_stream_wait_for_ctx": (_I, [_P, _P]),
    "sr3_set_precision": (_I, [_P, _I]),
    "sr3_conv_f8_supported": (_I, [_I, _I, _I, _I, _I]),
    "sr3_num_params": (_I, [_P]),
    "sr3_param_info": (_I, [_P, _I, C.c_char_p, _I, C.POINTER(C.c_int64), C.POINTER(_I)]),
    "sr3_load_weight": (_I, [_P, C.c_char_p, _F, C.POINTER(C.c_int64), _I]),
    "sr3_weights_missing": (_I, [_P]),
    "sr3_unet_forward": (_I, [_P, _F, _F, _I, _I, _I, _F]),
    "sr3_set_schedule": (_I, [_P, _I, _F, _F, _F, _F, _F, _F]),
    "sr3_sample": (_I, [_P, _F, _I, _I, _I, _F, _U64, _U64, _F, _F]),
    "sr3_num_frames": (_I, [_P]),
    "sr3_max_batch": (_I, [_P, _I, _I]),
    "sr3_sample_begin": (_I, [_P, _F, _I, _I, _I, _F, _U64, _U64]),
    "sr3_sample_step": (_I, [_P, _I, _F]),
    "sr3_sample_end": (_I, [_P, _F]),
    "sr3_range_check": (_I, [_P]),
    "sr3_set_range_policy": (_I, [_P, _I]),
    "sr3_fallback_calls": (_I, [_P]),
    "sr3_replay_calls": (_I, [_P]),
    "sr3_test_flag_address": (_P, [_P]),
    "sr3_last_warning": (C.c_char_p, []),
    "sr3_philox_normal": (_I, [_P, _U64, _U64, C.c_uint32, _I, _F]),
    "sr3_profile_enable": (_I, [_P, _I]),
    "sr3_profile_reset": (_I, [_P]),
    "sr3_profile_get": (_I, [_P, _I, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "sr3_profile_dump_csv": (_I, [_P, C.c_char_p]),
    "sr3_bench_conv": (_I, [_P] + [_I] * 13 + [C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "sr3_op_conv2d": (_I, [_P, _F, _I, _F, _I, _I, _I, _I, _F, _F, _I, _I, _I, _I, _F, _F, _I, _F, _F, _F]),
    "sr3_op_groupnorm_affine": (_I, [_P, _F, _I, _F, _I, _I, _I, _I, _I, _F, _F, _F, _F]),
    "sr3_op_attention": (_I, [_P, _F, _I, _I, _I, _F]),
    "sr3_op_noise_embed": (_I, [_P, _F, _I, _F, _F]),
    "sr3_chan_bias_total": (_I, [_P]),
    "sr3_op_nchw_to_nhwc": (_I, [_P, _F, _I, _I, _I, _I, _F]),
    "sr3_op_nhwc_to_nchw": (_I, [_P, _F, _I, _I, _I, _I, _F]),
    "sr3_preprocess_bicubic": (_I, [_P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "sr3_postprocess_u8": (_I, [_P, _F, _I, _I, _I, _I, _I, _P, _P, _F, _F]),
    "sr3_postprocess_tensor_blob": (_I, [_P, _F, _I, _I, _I, _I, _F]),
    "sr3_dev_malloc": (_I, [_P, _U64, C.POINTER(_P)]),
    "sr3_dev_free": (_I, [_P, _P]),
    "sr3_memcpy_h2d": (_I, [_P, _P, _P, _U64]),
    "sr3_memcpy_d2h": (_I, [_P, _P, _P, _U64]),
    "sr3_device_bytes": (_U64, [_P]),
}

_lib = None


def load() -> C.CDLL:
    """Loads libsr3hip.so and binds every prototype. Raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Sr3Error(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "There is no CPU fallback for the sampler."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)      # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


SR3_OK_F32_FALLBACK = 1
SR3_OK_REPLAYED = 2


def check(rc: int) -> None:
    """< 0: raise with the library's message; > 0 (SR3_OK_F32_FALLBACK, SR3_OK_REPLAYED): the result is valid, warn."""
    if rc == 0:
        return
    if rc > 0:
        import warnings
        msg = load().sr3_last_warning()
        kind = Sr3ReplayWarning if rc == SR3_OK_REPLAYED else Sr3RangeWarning
        warnings.warn(kind(msg.decode("utf-8", "replace") if msg else "recomputed"), stacklevel=3)
        return
    msg = load().sr3_last_error()
    raise Sr3Error(msg.decode("utf-8", "replace") if msg else f"libsr3hip error {rc}")
