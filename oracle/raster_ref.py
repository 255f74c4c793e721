"""ctypes front end of oracle/raster_ref.c (test infrastructure; parity unpinned — see the C header)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libraster_ref.so')


class RasterParams(C.Structure):
    _fields_ = [('quat', C.c_float * 4), ('trans', C.c_float * 3), ('bev_img_offset', C.c_float * 2),
                ('img_reso', C.c_float * 2), ('local_min_ele', C.c_float), ('ele_reso', C.c_float),
                ('inten_lo', C.c_float), ('inten_hi', C.c_float)]


def build():
    src = os.path.join(_HERE, 'raster_ref.c')
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s'])
    return _SO


def _lib():
    L = C.CDLL(build())
    L.raster_ref.argtypes = [C.c_void_p, C.c_long, C.POINTER(RasterParams), C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.pixel_to_point_ref.argtypes = [C.POINTER(RasterParams), C.c_double, C.c_double, C.c_double, C.c_void_p]
    return L


def params(quat=(1, 0, 0, 0), trans=(0, 0, 0), bev_img_offset=(0, 0), img_reso=(0.05, 0.05), local_min_ele=0.0,
           ele_reso=0.05, inten_lo=800.0, inten_hi=33000.0):
    p = RasterParams()
    p.quat[:] = [float(v) for v in quat]
    p.trans[:] = [float(v) for v in trans]
    p.bev_img_offset[:] = [float(v) for v in bev_img_offset]
    p.img_reso[:] = [float(v) for v in img_reso]
    p.local_min_ele, p.ele_reso, p.inten_lo, p.inten_hi = float(local_min_ele), float(ele_reso), float(inten_lo), float(inten_hi)
    return p


def raster(points, p, H=1152, W=1152):
    """points [N,4] f32 -> u8 HWC image [H,W,3] (R = B = brightest return, G = its elevation)."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    acc = np.zeros(H * W, dtype=np.uint32)
    out = np.zeros((H, W, 3), dtype=np.uint8)
    _lib().raster_ref(pts.ctypes.data, pts.shape[0], C.byref(p), acc.ctypes.data, out.ctypes.data, H, W)
    return out


def pixel_to_point(p, row, col, g):
    out = np.zeros(3, dtype=np.float64)
    _lib().pixel_to_point_ref(C.byref(p), float(row), float(col), float(g), out.ctypes.data)
    return out
