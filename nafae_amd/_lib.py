"""ctypes binding of libnafae_hip.so (include/nafae_hip.h).  There is NO fallback: if the library is missing or
does not load, every op raises -- the product path never routes through PyTorch eager ops or the CPU oracle."""
import ctypes
import os

# torch ships its own HIP runtime (torch/lib/libamdhip64.so).  It MUST be the one already resident when
# libnafae_hip.so is dlopen'ed, otherwise the process ends up with two runtimes and kernels launched through this
# library never touch torch's device memory.  Importing torch first makes the dynamic linker reuse its runtime.
import torch  # noqa: F401  (load order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
# NAFAE_LIB points at another build of the same library (kernel A/B experiments); the default is the in-tree build
NAFAE_OK, NAFAE_EINVAL, NAFAE_ELIMIT, NAFAE_ELAUNCH = 0, -1, -2, -3     # include/nafae_hip.h
LIB_PATH = os.environ.get("NAFAE_LIB") or os.path.join(_HERE, "csrc", "libnafae_hip.so")
_lib = None

c_int, c_float, c_void_p, c_int64 = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_int64
P = c_void_p

# name -> (restype, argtypes); must list every symbol declared in include/nafae_hip.h
SIGNATURES = {
    "nafae_version": (c_int, [ctypes.c_char_p, c_int]),
    "nafae_nms": (c_int, [P, P, P, c_int, c_int, c_float, P]),
    "nafae_roi_align_forward": (c_int, [c_int, c_int, c_float, P, c_int, c_int, c_int, c_int, P, c_int, P, P]),
    "nafae_roi_align_backward": (c_int, [c_int, c_int, c_float, P, P, c_int, P, c_int, c_int, c_int, c_int, P]),
    "nafae_gemm_nt": (c_int, [P, c_int, P, c_int, P, c_int, P, c_int, c_int, c_int, c_float, c_int, P]),
    "nafae_gemm_nt_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "nafae_gemm_nt_ws": (c_int, [P, c_int, P, c_int, P, c_int, P, c_int, c_int, c_int, c_float, c_int, P, c_int64, P]),
    "nafae_gemm_tn": (c_int, [P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_float, c_int, P]),
    "nafae_gemm_tn_rows": (c_int, [P, c_int, P, c_int, P, c_int, c_int, c_int, P, P, c_int, c_float, P]),
    "nafae_gemm_tn_rows_acc": (c_int, [P, c_int, P, c_int, P, c_int, c_int, c_int, P, P, c_int, c_float, c_int, P]),
    "nafae_nonzero_rows": (c_int, [P, c_int, c_int, P, P, P, P]),
    "nafae_conv1_3x3_relu": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "nafae_conv3x3_relu": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "nafae_conv3x3_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "nafae_conv3x3_relu_ws": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int64, P]),
    "nafae_conv3x3_wino_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "nafae_conv3x3_wino_weight_bytes": (c_int64, [c_int, c_int]),
    "nafae_conv3x3_wino_pack": (c_int, [P, P, c_int, c_int, P]),
    "nafae_conv3x3_wino": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "nafae_conv3x3_wino_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "nafae_conv3x3_wino_ws": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int64, P]),
    "nafae_maxpool2x2": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "nafae_rpn_decode": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "nafae_sort_desc": (c_int, [P, P, c_int, c_int, P]),
    "nafae_proposals": (c_int, [P, P, P, c_int, c_int, c_int, c_float, c_int, P, P, P, P]),
    "nafae_roi_align_avg_nhwc": (c_int, [P, c_int, c_int, c_int, c_int, P, c_int, c_float, P, P]),
    "nafae_frames_u8_to_nchw_f32": (c_int, [P, P, c_int, c_int, c_int, P]),
    "nafae_conv1_3x3_relu_in": (c_int, [P, c_int, P, P, P, c_int, c_int, c_int, P]),
    "nafae_conv1_3x3_relu_bf16_in": (c_int, [P, c_int, P, P, P, P, c_int, c_int, c_int, P]),
    "nafae_frames_resize_bilinear": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "nafae_nchw_to_nhwc": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "nafae_nhwc_to_nchw": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "nafae_split_bf16": (c_int, [P, P, P, c_int64, P]),
    "nafae_merge_bf16": (c_int, [P, P, P, c_int64, P]),
    "nafae_gemm_nt_bf16": (c_int, [P, P, c_int, P, P, c_int, P, P, P, c_int, P, c_int, c_int, c_int, c_float, c_int, P]),
    "nafae_conv3x3_bf16": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "nafae_conv3x3_bf16_ws": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, ctypes.c_int64, P]),
    "nafae_conv3x3_bf16_workspace_bytes": (ctypes.c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "nafae_conv1_3x3_relu_bf16": (c_int, [P, P, P, P, P, c_int, c_int, c_int, P]),
    "nafae_maxpool2x2_bf16": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "nafae_roi_align_avg_nhwc_bf16": (c_int, [P, P, c_int, c_int, c_int, c_int, P, c_int, c_float, P, P, P, P]),
    "nafae_roi_align_avg_nhwc_to_planes": (c_int, [P, c_int, c_int, c_int, c_int, P, c_int, c_float, P, P, P, P]),
    "nafae_sim_max_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "nafae_sim_max_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "nafae_sim_max_fwd_ws": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int64, P]),
    "nafae_sim_planes_bytes": (c_int64, [c_int, c_int, c_int]),
    "nafae_sim_planes": (c_int, [P, c_int, c_int, c_int, P, P, P]),
    "nafae_dropout_tanh_planes": (c_int, [P, P, c_float, P, c_int, c_int, c_int, P, P, P]),
    "nafae_dropout_tanh_seeded_planes": (c_int, [P, ctypes.c_uint64, c_float, P, c_int, c_int, c_int, P, P, P]),
    "nafae_sim_max_fwd_planes": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, c_int64, P]),
    "nafae_sim_planes_used": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "nafae_jpeg_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "nafae_jpeg_decode_batch": (c_int, [P, c_int64, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int64, P, P]),
    "nafae_loss_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "nafae_loss_fwd_bwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_int, P, P, P, P]),
    "nafae_loss_fwd_bwd_ex": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_int, c_int, P, P, P, P]),
    "nafae_sim_bwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P]),
    "nafae_sim_max_fwd_frames": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "nafae_sim_bwd_frames": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P]),
    "nafae_dropout_tanh": (c_int, [P, P, c_float, P, c_int64, P]),
    "nafae_dropout_tanh_bwd": (c_int, [P, P, P, c_float, P, c_int64, P]),
    "nafae_dropout_tanh_seeded": (c_int, [P, ctypes.c_uint64, c_float, P, c_int64, P]),
    "nafae_dropout_tanh_bwd_seeded": (c_int, [P, P, ctypes.c_uint64, c_float, P, c_int64, P]),
    "nafae_batchnorm_fwd": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_float, c_float, P]),
    "nafae_batchnorm_bwd": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, P]),
    "nafae_batchnorm_bwd_acc": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, P]),
    "nafae_colsum": (c_int, [P, P, c_int, c_int, P]),
    "nafae_colsum_acc": (c_int, [P, P, c_int, c_int, c_int, P]),
    "nafae_colsum_rows": (c_int, [P, P, P, c_int, c_int, P, c_int, P]),
    "nafae_adam_step": (c_int, [P, P, P, P, c_int64, c_float, c_float, c_float, c_float, c_float, c_float, c_int, P, P, P]),
}


class NafaeLibraryError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NafaeLibraryError(
                "libnafae_hip.so not built (%s). Run `python -m nafae_amd.build`; there is no fallback path." % LIB_PATH)
        try:
            l = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise NafaeLibraryError("cannot load %s: %s" % (LIB_PATH, e))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def version():
    buf = ctypes.create_string_buffer(64)
    rc = lib().nafae_version(buf, 64)
    if rc != 0:
        raise NafaeLibraryError("nafae_version failed: %d" % rc)
    return buf.value.decode()
