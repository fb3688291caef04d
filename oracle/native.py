"""ORACLE (test infrastructure): ctypes binding of oracle/native.c (gcc-built C restatement of the
reference's CUDA-only NMS and ROI-Align).  See native.c for file:line citations."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "native.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "_build/liboracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_iou.restype = ctypes.c_float
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def iou(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return float(lib().oracle_iou(_fp(a), _fp(b)))


def nms(dets, thresh):
    """dets [n, >=4] float32 sorted by descending score -> int32 keep positions (ascending)."""
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    n, dim = dets.shape
    keep = np.zeros(max(n, 1), dtype=np.int32)
    num = ctypes.c_int(0)
    lib().oracle_nms(keep.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ctypes.byref(num), _fp(dets),
                     ctypes.c_int(n), ctypes.c_int(dim), ctypes.c_float(thresh))
    return keep[:num.value].copy()


def roi_align_forward(features, rois, AH, AW, scale):
    """features [B,C,H,W], rois [N,5] -> [N,C,AH,AW] (roi_align_kernel.cu:15-70)."""
    features = np.ascontiguousarray(features, dtype=np.float32)
    rois = np.ascontiguousarray(rois, dtype=np.float32)
    B, C, H, W = features.shape
    N = rois.shape[0]
    out = np.zeros((N, C, AH, AW), dtype=np.float32)
    lib().oracle_roi_align_forward(_fp(features), ctypes.c_float(scale), N, H, W, C, AH, AW, _fp(rois), _fp(out))
    return out


def roi_align_backward(top_grad, rois, feature_shape, scale):
    """top_grad [N,C,AH,AW], rois [N,5] -> bottom_grad [B,C,H,W] (roi_align_kernel.cu:93-141; index-order adds)."""
    top_grad = np.ascontiguousarray(top_grad, dtype=np.float32)
    rois = np.ascontiguousarray(rois, dtype=np.float32)
    B, C, H, W = feature_shape
    N, C2, AH, AW = top_grad.shape
    assert C2 == C and rois.shape == (N, 5)
    out = np.zeros((B, C, H, W), dtype=np.float32)
    lib().oracle_roi_align_backward(_fp(top_grad), ctypes.c_float(scale), N, H, W, C, AH, AW, _fp(rois), _fp(out))
    return out


def roi_align_avg(features, rois, pooled, scale):
    """RoIAlignAvg (modules/roi_align.py:26-29): [N,C,pooled,pooled]."""
    features = np.ascontiguousarray(features, dtype=np.float32)
    rois = np.ascontiguousarray(rois, dtype=np.float32)
    B, C, H, W = features.shape
    N = rois.shape[0]
    out = np.zeros((N, C, pooled, pooled), dtype=np.float32)
    lib().oracle_roi_align_avg(_fp(features), ctypes.c_float(scale), N, H, W, C, pooled, _fp(rois), _fp(out))
    return out
