"""ctypes wrapper of oracle/lanms_oracle.c (test infrastructure only; see that file's header)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblanms_oracle.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", _HERE])
        _lib = ctypes.CDLL(_SO)
        _lib.lanms_oracle.restype = ctypes.c_int
        _lib.lanms_oracle_iou.restype = ctypes.c_float
    return _lib


def iou(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return float(_load().lanms_oracle_iou(a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p)))


def lanms(boxes, thr=0.2):
    """boxes [k,9] -> (merged [m,9], keep indices into merged in keep order)."""
    boxes = np.ascontiguousarray(boxes, np.float32)
    k = boxes.shape[0]
    merged = np.zeros((max(k, 1), 9), np.float32)
    keep = np.zeros(max(k, 1), np.int32)
    nm = ctypes.c_int(0)
    nk = _load().lanms_oracle(boxes.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(k), ctypes.c_float(thr),
                              merged.ctypes.data_as(ctypes.c_void_p), ctypes.byref(nm),
                              keep.ctypes.data_as(ctypes.c_void_p))
    return merged[:nm.value], keep[:nk]
