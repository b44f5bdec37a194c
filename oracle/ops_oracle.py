"""ctypes/numpy front-end of ``oracle/ops_oracle.c`` (the plain-C CPU restatement of the
reference's rotated-detection ops).  TEST INFRASTRUCTURE ONLY -- see the C file's header.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'ops_oracle.c')
LIB = os.path.join(HERE, 'libops_oracle.so')

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-shared', '-fPIC', SRC, '-o', LIB,
                               '-lm'])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.oracle_single_box_iou_rotated.restype = ctypes.c_float
        _lib.oracle_nms_rotated.restype = ctypes.c_int64
        _lib.oracle_nms.restype = ctypes.c_int64
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def box_iou_rotated(boxes1, boxes2, mode_flag=0, aligned=False):
    b1, p1 = _f(boxes1)
    b2, p2 = _f(boxes2)
    n, m = b1.shape[0], b2.shape[0]
    out = np.zeros(n if aligned else n * m, dtype=np.float32)
    if out.size:
        lib().oracle_box_iou_rotated(p1, n, p2, m, out.ctypes.data_as(_f32p), int(mode_flag),
                                     int(bool(aligned)))
    return out if aligned else out.reshape(n, m)


def nms_rotated(dets, scores, iou_threshold):
    d, pd = _f(dets)
    s, ps = _f(scores)
    n = d.shape[0]
    keep = np.zeros(max(n, 1), dtype=np.int64)
    k = lib().oracle_nms_rotated(pd, int(d.shape[1]) if n else 5, ps, ctypes.c_int64(n),
                                 ctypes.c_float(iou_threshold), keep.ctypes.data_as(_i64p))
    return keep[:k].copy()


def nms(boxes, scores, iou_threshold, offset=0):
    b, pb = _f(boxes)
    s, ps = _f(scores)
    n = b.shape[0]
    keep = np.zeros(max(n, 1), dtype=np.int64)
    k = lib().oracle_nms(pb, ps, ctypes.c_int64(n), ctypes.c_float(iou_threshold), int(offset),
                         keep.ctypes.data_as(_i64p))
    return keep[:k].copy()


def roi_align_rotated_forward(inp, rois, pooled_height, pooled_width, spatial_scale, sampling_ratio,
                              aligned=True, clockwise=False):
    x, px = _f(inp)
    r, pr = _f(rois)
    B, C, H, W = x.shape
    n = r.shape[0]
    out = np.zeros((n, C, pooled_height, pooled_width), dtype=np.float32)
    if n:
        lib().oracle_roi_align_rotated_forward(px, pr, out.ctypes.data_as(_f32p), n, C, H, W,
                                               pooled_height, pooled_width,
                                               ctypes.c_float(spatial_scale), int(sampling_ratio),
                                               int(bool(aligned)), int(bool(clockwise)))
    return out


def roi_align_rotated_backward(grad_output, rois, input_shape, pooled_height, pooled_width,
                               spatial_scale, sampling_ratio, aligned=True, clockwise=False):
    g, pg = _f(grad_output)
    r, pr = _f(rois)
    B, C, H, W = input_shape
    n = r.shape[0]
    gin = np.zeros((B, C, H, W), dtype=np.float32)
    if n:
        lib().oracle_roi_align_rotated_backward(pg, pr, gin.ctypes.data_as(_f32p), n, C, H, W,
                                                pooled_height, pooled_width,
                                                ctypes.c_float(spatial_scale), int(sampling_ratio),
                                                int(bool(aligned)), int(bool(clockwise)))
    return gin
