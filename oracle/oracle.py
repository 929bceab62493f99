"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY -- see lk_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product path (optical-flow_amd/) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_OMP = None

_dp = ctypes.POINTER(ctypes.c_double)
_fp = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    """Compile liblk_oracle.so / liblk_cpu_omp.so with gcc (a few seconds)."""
    targets = ["liblk_oracle.so", "liblk_cpu_omp.so"]
    if force or not all(os.path.exists(os.path.join(_HERE, t)) and
                        os.path.getmtime(os.path.join(_HERE, t)) >= max(os.path.getmtime(os.path.join(_HERE, s))
                                                                         for s in ("lk_oracle.c", "lk_cpu_omp.c", "lk_oracle.h"))
                        for t in targets):                  # (mtime test here: the GPU box may have no `make`-worthy clock skew)
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []) + targets)


def _p(a):
    return a.ctypes.data_as(_dp)


def _c64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    assert a.ndim == 2
    return a


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liblk_oracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.lko_max_layer.argtypes = [ctypes.c_int, ctypes.c_int]
        L.lko_max_layer.restype = ctypes.c_int
        for name in ("lko_get_fx", "lko_get_fy", "lko_get_ft"):
            getattr(L, name).argtypes = [_dp, _dp, ctypes.c_int, ctypes.c_int, _dp]
            getattr(L, name).restype = None
        L.lko_sum_win.argtypes = [_dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp]
        L.lko_sum_win.restype = None
        L.lko_lucas_kanade.argtypes = [_dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp]
        L.lko_lucas_kanade.restype = None
        L.lko_pyr_down.argtypes = [_dp, ctypes.c_int, ctypes.c_int, _dp]
        L.lko_pyr_down.restype = None
        L.lko_pyr_up.argtypes = [_dp, ctypes.c_int, ctypes.c_int, _dp]
        L.lko_pyr_up.restype = None
        L.lko_build_map.argtypes = [_dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_float, _fp]
        L.lko_build_map.restype = None
        L.lko_remap_linear.argtypes = [_dp, ctypes.c_int, ctypes.c_int, _fp, _dp]
        L.lko_remap_linear.restype = None
        L.lko_coarse_to_fine.argtypes = [_dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, _dp, _dp]
        L.lko_coarse_to_fine.restype = ctypes.c_int
        L.lko_u8_to_f64.argtypes = [ctypes.POINTER(ctypes.c_ubyte), ctypes.c_long, _dp]
        L.lko_u8_to_f64.restype = None
        L.lko_flow_centroid.argtypes = [_dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                        ctypes.POINTER(ctypes.c_longlong), _dp, _dp]
        L.lko_flow_centroid.restype = None
        L.lko_edge_map_u8.argtypes = [ctypes.POINTER(ctypes.c_ubyte), ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp]
        L.lko_edge_map_u8.restype = None
        _bp = ctypes.POINTER(ctypes.c_ubyte)
        L.lko_gaussian_kernel_q8.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_int)]
        L.lko_gaussian_kernel_q8.restype = None
        L.lko_blur_gray_bgr8.argtypes = [_bp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, _bp]
        L.lko_blur_gray_bgr8.restype = None
        L.lko_frame_diff_u8.argtypes = [_bp, _bp, ctypes.c_long, ctypes.c_double, _bp]
        L.lko_frame_diff_u8.restype = None
        L.lko_resize_cubic_tables.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_short)]
        L.lko_resize_cubic_tables.restype = None
        L.lko_resize_cubic_bgr8.argtypes = [_bp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _bp]
        L.lko_resize_cubic_bgr8.restype = None
        _LIB = L
    return _LIB


def max_layer(cols, rows):
    return lib().lko_max_layer(cols, rows)


def derivatives(i1, i2):
    i1, i2 = _c64(i1), _c64(i2)
    h, w = i1.shape
    out = [np.empty((h, w)) for _ in range(3)]
    for name, o in zip(("lko_get_fx", "lko_get_fy", "lko_get_ft"), out):
        getattr(lib(), name)(_p(i1), _p(i2), h, w, _p(o))
    return out


def sum_win(m, win=3):
    m = _c64(m)
    h, w = m.shape
    out = np.empty((h, w))
    lib().lko_sum_win(_p(m), h, w, win, _p(out))
    return out


def lucas_kanade(i1, i2, win=3):
    i1, i2 = _c64(i1), _c64(i2)
    h, w = i1.shape
    u, v = np.empty((h, w)), np.empty((h, w))
    lib().lko_lucas_kanade(_p(i1), _p(i2), h, w, win, _p(u), _p(v))
    return u, v


def pyr_down(a):
    a = _c64(a)
    h, w = a.shape
    out = np.empty(((h + 1) // 2, (w + 1) // 2))
    lib().lko_pyr_down(_p(a), h, w, _p(out))
    return out


def pyr_up(a):
    a = _c64(a)
    h, w = a.shape
    out = np.empty((2 * h, 2 * w))
    lib().lko_pyr_up(_p(a), h, w, _p(out))
    return out


def build_map(upu, upv, sign):
    upu, upv = _c64(upu), _c64(upv)
    h, w = upu.shape
    m = np.empty((h, w, 2), dtype=np.float32)
    lib().lko_build_map(_p(upu), _p(upv), h, w, float(sign), m.ctypes.data_as(_fp))
    return m


def remap_linear(src, map_xy):
    src = _c64(src)
    h, w = src.shape
    map_xy = np.ascontiguousarray(map_xy, dtype=np.float32)
    assert map_xy.shape == (h, w, 2)
    out = np.empty((h, w))
    lib().lko_remap_linear(_p(src), h, w, map_xy.ctypes.data_as(_fp), _p(out))
    return out


def coarse_to_fine(img1, img2, levels, win=3, mutate=False):
    """Returns (u, v).  With mutate=True img1/img2 (C-contiguous float64 arrays) are overwritten
    by the level-0 warp exactly as the reference overwrites its inputs; otherwise copies are used."""
    if mutate:
        assert img1.flags.c_contiguous and img1.dtype == np.float64
        assert img2.flags.c_contiguous and img2.dtype == np.float64
        a, b = img1, img2
    else:
        a, b = np.array(img1, dtype=np.float64, order="C"), np.array(img2, dtype=np.float64, order="C")
    h, w = a.shape
    u, v = np.empty((h, w)), np.empty((h, w))
    rc = lib().lko_coarse_to_fine(_p(a), _p(b), h, w, levels, win, _p(u), _p(v))
    if rc != 0:
        raise ValueError(f"lko_coarse_to_fine rc={rc}")
    return u, v


def u8_to_f64(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    out = np.empty(a.shape, dtype=np.float64)
    lib().lko_u8_to_f64(a.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), a.size, _p(out))
    return out


def edge_map_u8(diff, thresh=10):
    diff = np.ascontiguousarray(diff, dtype=np.uint8)
    assert diff.ndim == 2
    out = np.empty(diff.shape, dtype=np.float64)
    lib().lko_edge_map_u8(diff.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), diff.shape[0], diff.shape[1],
                          int(thresh), _p(out))
    return out


def gaussian_kernel_q8(ksize=9, sigma=1.5):
    k = (ctypes.c_int * ksize)()
    lib().lko_gaussian_kernel_q8(ksize, float(sigma), k)
    return np.array(list(k), dtype=np.int64)


def blur_gray_bgr8(bgr, ksize=9, sigma=1.5):
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    assert bgr.ndim == 3 and bgr.shape[2] == 3
    out = np.empty(bgr.shape[:2], dtype=np.uint8)
    _bp = ctypes.POINTER(ctypes.c_ubyte)
    lib().lko_blur_gray_bgr8(bgr.ctypes.data_as(_bp), bgr.shape[0], bgr.shape[1], ksize, float(sigma),
                             out.ctypes.data_as(_bp))
    return out


def resize_cubic_tables(ssize, dsize):
    ofs = np.empty(dsize, dtype=np.int32)
    coef = np.empty((dsize, 4), dtype=np.int16)
    lib().lko_resize_cubic_tables(int(ssize), int(dsize), ofs.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                  coef.ctypes.data_as(ctypes.POINTER(ctypes.c_short)))
    return ofs, coef


def resize_cubic_bgr8(bgr, drows, dcols):
    """ParallelVideoPyr.cpp:778 resize(.., Size(dcols, drows), 0, 0, INTER_CUBIC) on an 8UC3 frame (parity unpinned)."""
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    assert bgr.ndim == 3 and bgr.shape[2] == 3
    out = np.empty((drows, dcols, 3), dtype=np.uint8)
    _bp = ctypes.POINTER(ctypes.c_ubyte)
    lib().lko_resize_cubic_bgr8(bgr.ctypes.data_as(_bp), bgr.shape[0], bgr.shape[1], int(drows), int(dcols),
                                out.ctypes.data_as(_bp))
    return out


def frame_diff_u8(cur, prev, rate=0.3):
    cur, prev = np.ascontiguousarray(cur, dtype=np.uint8), np.ascontiguousarray(prev, dtype=np.uint8)
    assert cur.shape == prev.shape
    out = np.empty(cur.shape, dtype=np.uint8)
    _bp = ctypes.POINTER(ctypes.c_ubyte)
    lib().lko_frame_diff_u8(cur.ctypes.data_as(_bp), prev.ctypes.data_as(_bp), cur.size, float(rate),
                            out.ctypes.data_as(_bp))
    return out


def flow_centroid(u, v, thresh=20.0):
    u, v = _c64(u), _c64(v)
    h, w = u.shape
    c = ctypes.c_longlong(0)
    sx, sy = ctypes.c_double(0), ctypes.c_double(0)
    lib().lko_flow_centroid(_p(u), _p(v), h, w, thresh, ctypes.byref(c), ctypes.byref(sx),
                            ctypes.byref(sy))
    return c.value, sx.value, sy.value


# ---- OpenMP restatement of ParallelVideoPyr.cpp's parallel structure (CPU baseline) ----------
def omp_lib():
    global _OMP
    if _OMP is None:
        path = os.path.join(_HERE, "liblk_cpu_omp.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.lkc_coarse_to_fine.argtypes = [_dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, _dp, _dp]
        L.lkc_coarse_to_fine.restype = ctypes.c_int
        L.lkc_num_threads.argtypes = []
        L.lkc_num_threads.restype = ctypes.c_int
        L.lkc_set_num_threads.argtypes = [ctypes.c_int]
        L.lkc_set_num_threads.restype = None
        _OMP = L
    return _OMP


def omp_coarse_to_fine(img1, img2, levels, win=3, mode=0):
    """mode 0 = the reference's task structure (2/3/5-way sections, 8-thread row loops);
    mode 1 = every stage row-parallel over all cores.  Inputs are copied (not mutated)."""
    a, b = np.array(img1, dtype=np.float64, order="C"), np.array(img2, dtype=np.float64, order="C")
    h, w = a.shape
    u, v = np.empty((h, w)), np.empty((h, w))
    rc = omp_lib().lkc_coarse_to_fine(_p(a), _p(b), h, w, levels, win, mode, _p(u), _p(v))
    if rc != 0:
        raise ValueError(f"lkc_coarse_to_fine rc={rc}")
    return u, v
