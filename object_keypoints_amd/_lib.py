"""ctypes binding of libokp_hip.so (the C ABI declared in include/okp.h).

There is no CPU fallback: if the shared library is missing or does not load, importing a
compute entry point raises.  Build it with `python -m object_keypoints_amd.build`
(or `__graft_entry__.build()`).
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p)

LIB_PATH = os.environ.get("OKP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libokp_hip.so")   # OKP_LIB: A/B builds

OKP_F32, OKP_BF16, OKP_F16, OKP_F32X3 = 0, 1, 2, 3
OKP_ABI = 7
CAM_EQUIDISTANT, CAM_RADTAN = 0, 1
ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2
HEAD_MAX_OUT = 32


class OkpError(RuntimeError):
    pass


class okp_tap(Structure):
    _fields_ = [("src", c_int32), ("dy", c_int32), ("dx", c_int32), ("w", POINTER(c_float))]


class okp_tensor(Structure):
    _fields_ = [("data", c_void_p), ("h", c_int32), ("w", c_int32), ("pix_stride", c_int32), ("bytes", c_int64)]


class okp_conv_args(Structure):
    _fields_ = [("n", c_int32), ("ho", c_int32), ("wo", c_int32),
                ("src", okp_tensor * 2), ("out", okp_tensor),
                ("out_step", c_int32), ("out_oy", c_int32), ("out_ox", c_int32),
                ("res", okp_tensor), ("tile", c_int32),
                ("dw_w_dev", c_void_p), ("dw_bias_dev", c_void_p), ("dw_out", okp_tensor), ("dw_res", okp_tensor),
                ("n_classes", c_int32), ("out16", okp_tensor), ("res_is_f16", c_int32), ("out_subsample", c_int32), ("src_pairs", c_int32), ("out_pairs", c_int32)]


class okp_fire_args(Structure):
    _fields_ = [("n", c_int32), ("x", okp_tensor), ("out", okp_tensor), ("stride", c_int32), ("skip", c_int32)]


class okp_head_out_args(Structure):
    _fields_ = [("n", c_int32), ("h", c_int32), ("w", c_int32), ("src", okp_tensor), ("n_out", c_int32),
                ("in_c_off", c_int32 * HEAD_MAX_OUT), ("act", c_int32 * HEAD_MAX_OUT),
                ("out_ptr", c_void_p * HEAD_MAX_OUT), ("out_n_stride", c_int64 * HEAD_MAX_OUT),
                ("w_dev", c_void_p), ("bias_dev", c_void_p)]


class okp_camera(Structure):
    _fields_ = [("fx", c_double), ("fy", c_double), ("cx", c_double), ("cy", c_double), ("d", c_double * 4),
                ("model", c_int32), ("reserved", c_int32)]


# every symbol include/okp.h declares: (name, restype, argtypes)
SIGNATURES = [
    ("okp_last_error", c_char_p, []),
    ("okp_abi_version", c_int, []),
    ("okp_device_count", c_int, []),
    ("okp_device_arch", c_int, [c_int, c_char_p, c_int]),
    ("okp_conv_create", c_void_p, [c_int, c_int, POINTER(c_int32), POINTER(c_int32), c_int32, c_int32, POINTER(okp_tap), POINTER(c_float), c_int]),
    ("okp_conv_create_x3", c_void_p, [c_int, POINTER(c_int32), POINTER(c_int32), c_int32, c_int32, POINTER(okp_tap), POINTER(ctypes.c_uint8), POINTER(c_float), c_int]),
    ("okp_conv_set_range_flag", c_int, [c_void_p, c_void_p]),
    ("okp_stem_set_range_flag", c_int, [c_void_p, c_void_p]),
    ("okp_conv_destroy", None, [c_void_p]),
    ("okp_cast", c_int, [c_int, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p]),
    ("okp_add_f16_f32", c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    ("okp_stream_wait_stream", c_int, [c_void_p, c_void_p]),
    ("okp_conv_forward", c_int, [c_void_p, POINTER(okp_conv_args), c_void_p]),
    ("okp_conv_select_tile", c_int, [c_void_p, POINTER(okp_conv_args)]),
    ("okp_conv_macs", c_int64, [c_void_p, POINTER(okp_conv_args)]),
    ("okp_conv_patch_applies", c_int, [c_void_p, POINTER(okp_conv_args)]),
    ("okp_fire_forward", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, POINTER(okp_fire_args), c_void_p]),
    ("okp_fire_chain_forward", c_int, [c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), c_int32,
                                       POINTER(okp_tensor), POINTER(okp_tensor), c_void_p]),
    ("okp_dwconv3x3_forward", c_int, [c_int, c_int32, c_int32, c_int32, POINTER(okp_tensor), c_void_p, c_void_p, POINTER(okp_tensor), POINTER(okp_tensor), c_int, c_void_p]),
    ("okp_pack_frames", c_int, [c_int, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    ("okp_pack_frames_u8", c_int, [c_int, c_void_p, c_int32, c_int32, c_int32, POINTER(c_float), POINTER(c_float), c_void_p, c_int32, c_void_p]),
    ("okp_preprocess_u8", c_int, [c_int, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                  POINTER(c_float), POINTER(c_float), c_void_p, c_int32, c_void_p]),
    ("okp_stem_create", c_void_p, [POINTER(c_float), POINTER(c_float)]),
    ("okp_stem_create_dtype", c_void_p, [c_int, POINTER(c_float), POINTER(c_float)]),
    ("okp_stem_destroy", None, [c_void_p]),
    ("okp_stem_forward", c_int, [c_void_p, c_int32, c_int32, c_int32, POINTER(okp_tensor), POINTER(okp_tensor), c_void_p]),
    ("okp_stem_forward_nchw", c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p, POINTER(okp_tensor), c_void_p]),
    ("okp_stem_forward_nchw_pairs", c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p, POINTER(okp_tensor), c_void_p]),
    ("okp_head_out_forward", c_int, [c_int, POINTER(okp_head_out_args), c_void_p]),
    ("okp_heads_forward", c_int, [c_void_p, c_void_p, POINTER(okp_head_out_args), POINTER(okp_tensor), c_void_p]),
    ("okp_peak_nms", c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("okp_nms_maxpool", c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    ("okp_unproject_depth", c_int, [POINTER(okp_camera), c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    ("okp_lift_peaks", c_int, [POINTER(okp_camera), c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    ("okp_group_objects", c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, POINTER(c_int32), c_float, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("okp_capacity_overflow", c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    ("okp_triangulate_dlt", c_int, [POINTER(okp_camera), POINTER(okp_camera), POINTER(c_double), POINTER(c_double), c_int, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    ("okp_fisheye_undistort", c_int, [POINTER(okp_camera), c_void_p, c_int32, c_void_p, c_void_p]),
    ("okp_camera_undistort", c_int, [POINTER(okp_camera), c_void_p, c_int32, c_void_p, c_void_p]),
]

_lib = None


def lib():
    """The loaded library; raises OkpError (never falls back) when it is unavailable."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OkpError(f"{LIB_PATH} not found: build it with `python -m object_keypoints_amd.build`; "
                           "there is no CPU fallback for the HIP path")
        try:
            # One HIP runtime per process: torch ships its own libamdhip64 (same SONAME as /opt/rocm's).  Load
            # torch's copy first so that libokp_hip.so binds to the runtime that owns torch's streams and
            # allocations; two runtime instances do not share devices ("no ROCm-capable device is detected").
            import torch
            bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
            if os.path.exists(bundled):
                ctypes.CDLL(bundled, mode=ctypes.RTLD_GLOBAL)
            handle = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise OkpError(f"cannot load {LIB_PATH}: {e}") from e
        for name, restype, argtypes in SIGNATURES:
            fn = getattr(handle, name)          # AttributeError = ABI mismatch, let it surface
            fn.restype = restype
            fn.argtypes = argtypes
        if handle.okp_abi_version() != OKP_ABI:
            raise OkpError("libokp_hip.so ABI version mismatch")
        _lib = handle
    return _lib


TORCH_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libokp_torch.so")
_torch_ops = False


def torch_ops():
    """torch.ops.okp (csrc/okp_torch.cpp: the launch entry points registered with the PyTorch dispatcher, a shim over the same
    C ABI) or None when the shim is not built or switched off (OKP_TORCH_OPS=0); the ctypes binding then serves the launches."""
    global _torch_ops
    if _torch_ops is False:
        _torch_ops = None
        if os.environ.get("OKP_TORCH_OPS", "1") != "0" and os.path.exists(TORCH_LIB_PATH) and not os.environ.get("OKP_LIB"):
            lib()                                   # libokp_hip.so first (and torch's HIP runtime before it)
            import torch
            try:
                torch.ops.load_library(TORCH_LIB_PATH)
                _torch_ops = torch.ops.okp
            except (OSError, RuntimeError) as e:    # a shim built against another torch: fall back to ctypes, loudly
                import warnings
                warnings.warn(f"{TORCH_LIB_PATH} does not load ({e}); using the ctypes binding of libokp_hip.so")
    return _torch_ops


def check(rc, what=""):
    if rc != 0:
        msg = lib().okp_last_error()
        raise OkpError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
