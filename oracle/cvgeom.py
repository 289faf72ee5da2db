"""ctypes wrapper of oracle/cvgeom_oracle.c (test infrastructure only; see that file's header)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcvgeom_oracle.so")
_SRC = os.path.join(_HERE, "cvgeom_oracle.c")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or (os.path.exists(_SRC) and os.path.getmtime(_SRC) > os.path.getmtime(_SO)):
            subprocess.check_call(["make", "-C", _HERE])
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def convex_hull(points):
    """cv2.convexHull(points int32 [n,2], clockwise=True) -> [m,2] int32."""
    pts = np.ascontiguousarray(points, np.int32).reshape(-1, 2)
    out = np.zeros((max(len(pts), 1), 2), np.int32)
    m = _load().cvgeom_convex_hull(_p(pts), ctypes.c_int(len(pts)), _p(out))
    return out[:m]


def min_area_rect(points):
    """cv2.minAreaRect(points int [n,2]) -> ((cx,cy),(w,h),angle) as a float32 [5], plus the
    calipers' raw (corner, edge1, edge2) float32 [6] and the hull."""
    pts = np.ascontiguousarray(points, np.int32).reshape(-1, 2)
    rect = np.zeros(5, np.float32)
    cal = np.zeros(6, np.float32)
    hull = np.zeros((max(len(pts), 1), 2), np.int32)
    nh = _load().cvgeom_min_area_rect(_p(pts), ctypes.c_int(len(pts)), _p(rect), _p(cal), _p(hull))
    return rect, cal, hull[:nh]


def rect_from_hull(hull, cal6):
    hull = np.ascontiguousarray(hull, np.int32).reshape(-1, 2)
    cal6 = np.ascontiguousarray(cal6, np.float32)
    rect = np.zeros(5, np.float32)
    _load().cvgeom_rect_from_hull(_p(hull), ctypes.c_int(len(hull)), _p(cal6), _p(rect))
    return rect


def box_points(rect):
    """cv2.boxPoints(((cx,cy),(w,h),angle)) -> float32 [4,2]."""
    rect = np.ascontiguousarray(rect, np.float32)
    pts = np.zeros((4, 2), np.float32)
    _load().cvgeom_box_points(_p(rect), _p(pts))
    return pts


def fill_poly(img, pts, color):
    """cv2.fillPoly(img uint8 [h,w], [pts int32 [k,2]], color) in place."""
    assert img.dtype == np.uint8 and img.ndim == 2 and img.flags["C_CONTIGUOUS"]
    pts = np.ascontiguousarray(pts, np.int32).reshape(-1, 2)
    _load().cvgeom_fill_poly(_p(img), ctypes.c_int(img.shape[0]), ctypes.c_int(img.shape[1]), _p(pts),
                             ctypes.c_int(len(pts)), ctypes.c_int(int(color)))
    return img


def resize_cubic_f32(src, dh, dw):
    """cv2.resize(src float32 [h,w], (dw, dh), interpolation=cv2.INTER_CUBIC)."""
    src = np.ascontiguousarray(src, np.float32)
    h, w = src.shape
    dst = np.zeros((dh, dw), np.float32)
    _load().cvgeom_resize_cubic_f32(_p(src), ctypes.c_int(h), ctypes.c_int(w), _p(dst), ctypes.c_int(dh),
                                    ctypes.c_int(dw))
    return dst


def resize_linear_u8(src, dh, dw):
    """cv2.resize(src uint8 [H,W,cn], dsize=(dw, dh)) with the default INTER_LINEAR."""
    src = np.ascontiguousarray(src, np.uint8)
    if src.ndim == 2:
        src = src[:, :, None]
    H, W, cn = src.shape
    dst = np.zeros((dh, dw, cn), np.uint8)
    _load().cvgeom_resize_linear_u8(_p(src), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(cn), _p(dst),
                                    ctypes.c_int(dh), ctypes.c_int(dw))
    return dst
