"""TEST INFRASTRUCTURE - ctypes wrapper of oracle/nms_ref.c (see that file's header). Not product code."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libnmsref.so')
_lib = None


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        h = C.CDLL(_SO)
        h.y5ref_nms.restype = C.c_int64
        h.y5ref_nms.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_int64, C.c_int]
        h.y5ref_nms_image.restype = C.c_int64
        h.y5ref_nms_image.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_float, C.c_double, C.c_void_p, C.c_int,
                                      C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = h
    return _lib


def nms(boxes, scores, thr, max_keep=None):
    """Kept indices (int64, descending score) of the C restatement of torchvision.ops.boxes.nms."""
    boxes = np.ascontiguousarray(boxes, np.float32)
    scores = np.ascontiguousarray(scores, np.float32)
    n = boxes.shape[0]
    cap = n if max_keep is None else max_keep
    keep = np.zeros(max(cap, 1), np.int64)
    k = lib().y5ref_nms(boxes.ctypes.data, scores.ctypes.data, n, float(thr), keep.ctypes.data, cap, 0 if max_keep is None else 1)
    assert k >= 0
    return keep[:min(k, cap)]


def non_max_suppression(prediction, conf_thres=0.1, iou_thres=0.6, classes=None, agnostic=False, max_det=300):
    """Per image (det (n,6) float32, kept flat index (n,), n_candidates) from the C stage list."""
    pred = np.ascontiguousarray(np.asarray(prediction).astype(np.float32))
    B, N, no = pred.shape
    mask = None
    if classes:
        mask = np.zeros(no - 5, np.uint8)
        mask[[int(c) for c in classes if 0 <= int(c) < no - 5]] = 1
    out = []
    for b in range(B):
        det = np.zeros((max_det, 6), np.float32)
        kept = np.zeros(max_det, np.int64)
        ncand = C.c_int64(0)
        k = lib().y5ref_nms_image(pred[b].ctypes.data, N, no, float(conf_thres), float(iou_thres),
                                  mask.ctypes.data if mask is not None else None, 1 if agnostic else 0, max_det,
                                  det.ctypes.data, kept.ctypes.data, C.byref(ncand))
        assert k >= 0
        out.append((det[:k].copy(), kept[:k].copy(), int(ncand.value)))
    return out
