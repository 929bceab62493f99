"""lkflow_amd -- Python host mirror of the reference's hot-call interface, over the C ABI.

The product is liblkflow.so (HIP, gfx950); this module is a thin ctypes binding used by the parity
tests, bench.py and the multi-GPU frame-pair sharding.  It mirrors the reference's own names:

    getMaxLayer(img)                                   ParallelVideoPyr.cpp:545
    getLucasKanadeOpticalFlow(img1, img2) -> (u, v)    ParallelVideoPyr.cpp:327
    coarseToFineEstimation(img1, img2, nLevels)        ParallelVideoPyr.cpp:578
    computeFlow(frame_prev, frame_next, pyr_levels, win)   (BASELINE.json's name for the same call)

torch is used only as plumbing: device memory (tensor.data_ptr()) and the current HIP stream.
There is NO CPU fallback: if liblkflow.so or a GPU is missing, calls raise.
"""
import ctypes
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LKFLOW_LIB") or os.path.join(_HERE, "liblkflow.so")   # LKFLOW_LIB: A/B builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "lkflow.h")

LKFLOW_OK = 0
LKFLOW_E_ARG = -1
LKFLOW_E_LEVELS = -2
LKFLOW_E_WIN = -3
LKFLOW_E_HIP = -4
LKFLOW_E_NOMEM = -5
LKFLOW_E_NODEVICE = -6
LKFLOW_MUTATE_INPUTS = 1
LKFLOW_STREAM_KEEP_FLOW = 1
LKFLOW_MAX_WIN = 31
LKFLOW_OPT_IMPL = 1
LKFLOW_OPT_TILE_ROWS = 2
LKFLOW_OPT_PROFILE = 3
LKFLOW_OPT_PIN_HOST = 4
LKFLOW_OPT_FUSE_WARP = 5
LKFLOW_OPT_LK_BATCH = 6
LKFLOW_STAGE_PYRDOWN, LKFLOW_STAGE_WARP, LKFLOW_STAGE_LK = 0, 1, 2
LKFLOW_N_STAGES, LKFLOW_MAX_LEVELS = 3, 32

_vp = ctypes.c_void_p
_sz = ctypes.c_size_t
_i = ctypes.c_int
_u = ctypes.c_uint
_d = ctypes.c_double

# name -> (restype, argtypes); must list EVERY function declared in include/lkflow.h
SIGNATURES = {
    "lkflow_max_levels": (_i, [_i, _i]),
    "lkflow_strerror": (ctypes.c_char_p, [_i]),
    "lkflow_last_error": (ctypes.c_char_p, [_vp]),
    "lkflow_version": (ctypes.c_char_p, []),
    "lkflow_lk_kernel_name": (_i, [_i, _i, _i, _i, _i, _i, ctypes.c_char_p, ctypes.c_size_t]),
    "lkflow_strip_plan": (_i, [_i, _i, ctypes.c_longlong, _i, _i, ctypes.POINTER(ctypes.c_longlong)]),
    "lkflow_ctx_create": (_i, [ctypes.POINTER(_vp), _i]),
    "lkflow_ctx_destroy": (None, [_vp]),
    "lkflow_ctx_reserve": (_i, [_vp, _i, _i, _i, _i]),
    "lkflow_ctx_set_option": (_i, [_vp, _i, _i]),
    "lkflow_ctx_workspace_bytes": (_sz, [_vp]),
    "lkflow_ctx_profile_read": (_i, [_vp, ctypes.POINTER(_d), ctypes.POINTER(_i)]),
    "lkflow_compute_f64": (_i, [_vp, _vp, _vp, _i, _i, _sz, _vp, _vp, _sz, _i, _i, _u]),
    "lkflow_compute_u8": (_i, [_vp, _vp, _vp, _i, _i, _sz, _vp, _vp, _sz, _i, _i]),
    "lkflow_enqueue_pair_f64": (_i, [_vp, _vp, _vp, _i, _i, _sz, _vp, _vp, _sz, _i, _i, _u]),
    "lkflow_enqueue_pair_u8": (_i, [_vp, _vp, _vp, _i, _i, _sz, _vp, _vp, _sz, _i, _i]),
    "lkflow_wait": (_i, [_vp]),
    "lkflow_lk_f64": (_i, [_vp, _vp, _vp, _i, _i, _sz, _vp, _vp, _sz, _i]),
    "lkflow_compute_batch_f64_dev": (_i, [_vp, _vp, _vp, _sz, _sz, _i, _i, _i, _sz, _vp, _vp, _sz,
                                          _sz, _i, _i, _u, _vp]),
    "lkflow_lk_batch_f64_dev": (_i, [_vp, _vp, _vp, _sz, _sz, _i, _i, _i, _sz, _vp, _vp, _sz, _sz, _i,
                                     _vp]),
    "lkflow_lk_up_batch_f64_dev": (_i, [_vp, _vp, _vp, _sz, _sz, _i, _i, _i, _sz, _vp, _vp, _sz, _sz, _vp, _vp, _sz, _sz,
                                        _i, _vp]),
    "lkflow_pyrdown_f64_dev": (_i, [_vp, _vp, _sz, _i, _i, _i, _sz, _vp, _sz, _sz, _vp]),
    "lkflow_pyrup_f64_dev": (_i, [_vp, _vp, _sz, _i, _i, _i, _sz, _vp, _sz, _sz, _vp]),
    "lkflow_pyrdown2_f64_dev": (_i, [_vp, _vp, _sz, _i, _i, _i, _sz, _vp, _sz, _sz, _vp, _sz, _sz, _vp]),
    "lkflow_warp_f64_dev": (_i, [_vp, _vp, _vp, _sz, _i, _i, _i, _sz, _vp, _vp, _sz, _sz, _vp, _vp,
                                 _sz, _sz, _vp]),
    "lkflow_u8_to_f64_dev": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "lkflow_solve_f64_dev": (_i, [_vp, _vp, _sz, _vp, _vp, _i, _vp]),
    "lkflow_blur_gray_bgr8_dev": (_i, [_vp, _vp, _sz, _i, _i, _i, _sz, _i, _d, _vp, _sz, _sz, _vp]),
    "lkflow_resize_cubic_bgr8_dev": (_i, [_vp, _vp, _sz, _i, _i, _i, _sz, _vp, _sz, _i, _i, _sz, _vp]),
    "lkflow_frame_diff_u8_dev": (_i, [_vp, _vp, _vp, _sz, _i, _i, _i, _sz, _d, _vp, _sz, _sz, _vp]),
    "lkflow_edge_map_u8_dev": (_i, [_vp, _vp, _sz, _i, _i, _i, _sz, _i, _vp, _sz, _sz, _vp]),
    "lkflow_diff_edge_map_u8_dev": (_i, [_vp, _vp, _vp, _sz, _i, _i, _i, _sz, _d, _i, _vp, _sz, _sz, _vp]),
    "lkflow_flow_centroid_f64_dev": (_i, [_vp, _vp, _vp, _sz, _i, _i, _i, _sz, _d, _vp, _vp]),
    "lkflow_stream_create": (_i, [ctypes.POINTER(_vp), ctypes.POINTER(_i), _i, _i, _i, _i, _i, _u]),
    "lkflow_stream_destroy": (None, [_vp]),
    "lkflow_stream_set_budget": (_i, [_vp, _sz]),
    "lkflow_stream_push_u8": (_i, [_vp, _vp, _i, _sz, _sz, _d]),
    "lkflow_stream_centroids": (_i, [_vp, _vp, _i]),
    "lkflow_stream_flow": (_i, [_vp, _i, _vp, _vp, _sz]),
    "lkflow_stream_block": (_i, [_i, _i, _i, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "lkflow_stream_last_error": (ctypes.c_char_p, [_vp]),
}

_LIB = None


class LKFlowError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = f"lkflow status {status}"
        try:
            msg += ": " + lib().lkflow_strerror(status).decode()
        except Exception:
            pass
        if detail:
            msg += f" [{detail}]"
        super().__init__(msg)


def header_functions():
    """Names of all functions declared in include/lkflow.h (for the ABI-completeness test)."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lkflow_[a-z0-9_]+)\s*\(", text)))


def lib():
    """Load liblkflow.so (raises if it has not been built -- there is no fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950); lkflow has no CPU fallback")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def lk_kernel_name(rows, cols, win, has_up=True, fused_ok=True, pairs=1):
    """Name (as rocprofv3 prints it) of the LK kernel the library launches for one pyramid level of a batch of `pairs`."""
    buf = ctypes.create_string_buffer(128)
    rc = lib().lkflow_lk_kernel_name(int(rows), int(cols), int(win), int(has_up), int(fused_ok), int(pairs), buf, 128)
    if rc != 0:
        raise LKFlowError(rc)
    return buf.value.decode()


def strip_plan(rows, halo_rows, column_strips, slots, lone_slots=0):
    """(rows per strip of phase A, column strips in phase A, rows per strip of phase B) of the wave-specialised LK launch."""
    out = (ctypes.c_longlong * 3)()
    rc = lib().lkflow_strip_plan(int(rows), int(halo_rows), int(column_strips), int(slots), int(lone_slots), out)
    if rc != 0:
        raise LKFlowError(rc)
    return int(out[0]), int(out[1]), int(out[2])


def getMaxLayer(img_or_cols, rows=None):
    """getMaxLayer (ParallelVideoPyr.cpp:545-564).  Accepts an image (H x W) or (cols, rows)."""
    if rows is None:
        h, w = img_or_cols.shape[-2:]
        return lib().lkflow_max_levels(int(w), int(h))
    return lib().lkflow_max_levels(int(img_or_cols), int(rows))


def _torch():
    import torch
    return torch


def _stream_ptr(stream=None):
    torch = _torch()
    s = stream if stream is not None else torch.cuda.current_stream()
    return ctypes.c_void_p(s.cuda_stream)


def _check_dev(t, name):
    torch = _torch()
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64):
        raise TypeError(f"{name}: expected a float64 CUDA(HIP) tensor")
    if t.stride(-1) != 1:
        raise ValueError(f"{name}: innermost stride must be 1")


class Context:
    """One lkflow_ctx (device workspace) on one GPU."""

    def __init__(self, device=0):
        self._h = _vp()
        rc = lib().lkflow_ctx_create(ctypes.byref(self._h), int(device))
        if rc != 0:
            self._h = None
            raise LKFlowError(rc)
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            lib().lkflow_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise LKFlowError(rc, lib().lkflow_last_error(self._h).decode())

    def set_option(self, key, value):
        self._chk(lib().lkflow_ctx_set_option(self._h, key, value))

    def reserve(self, rows, cols, levels, n_pairs=1):
        self._chk(lib().lkflow_ctx_reserve(self._h, rows, cols, levels, n_pairs))

    def profile(self, stage=None, level=0):
        """Bracket kernel launches with HIP events: profile() = all stages, profile(stage, level) =
        one, profile(False) = off."""
        if stage is False:
            v = 0
        elif stage is None:
            v = -1
        else:
            v = 1 + stage * LKFLOW_MAX_LEVELS + level
        self.set_option(LKFLOW_OPT_PROFILE, v)

    def profile_read(self):
        """-> {(stage, level): (total_ms, n_launches)} for everything recorded since the last read."""
        n = LKFLOW_N_STAGES * LKFLOW_MAX_LEVELS
        ms = (ctypes.c_double * n)()
        cnt = (ctypes.c_int * n)()
        self._chk(lib().lkflow_ctx_profile_read(self._h, ms, cnt))
        return {(i // LKFLOW_MAX_LEVELS, i % LKFLOW_MAX_LEVELS): (ms[i], cnt[i]) for i in range(n) if cnt[i]}

    @property
    def workspace_bytes(self):
        return lib().lkflow_ctx_workspace_bytes(self._h)

    # ---- host (numpy) drop-ins --------------------------------------------------------------
    def coarseToFineEstimation(self, img1, img2, nLevels, win=3, mutate=True):
        """coarseToFineEstimation(img1, img2, u, v, nLevels) on host arrays; returns (u, v).
        mutate=True reproduces the reference's in-place overwrite of img1/img2 (they must then be
        writable float64 arrays with unit inner stride)."""
        a, b = self._host_in(img1, mutate), self._host_in(img2, mutate)
        h, w = a.shape
        u, v = np.empty((h, w)), np.empty((h, w))
        rc = lib().lkflow_compute_f64(self._h, a.ctypes.data, b.ctypes.data, h, w, a.strides[0],
                                      u.ctypes.data, v.ctypes.data, u.strides[0], int(nLevels),
                                      int(win), LKFLOW_MUTATE_INPUTS if mutate else 0)
        self._chk(rc)
        return u, v

    def computeFlow(self, frame_prev, frame_next, pyr_levels, win=3):
        """BASELINE.json's computeFlow(frame_prev, frame_next, U, V, pyr_levels, win): same call,
        window generalised, inputs NOT mutated."""
        return self.coarseToFineEstimation(frame_prev, frame_next, pyr_levels, win, mutate=False)

    def computeFlow_u8(self, frame_prev, frame_next, pyr_levels, win=3):
        """computeFlow for 8-bit frames (what SerialPyr.cpp:223-238 loads): convertTo(CV_64F, 1/255) runs on the
        device, so 1 byte per pixel crosses PCIe.  Returns (U, V)."""
        a, b = self._host_u8(frame_prev), self._host_u8(frame_next)
        h, w = a.shape
        u, v = np.empty((h, w)), np.empty((h, w))
        self._chk(lib().lkflow_compute_u8(self._h, a.ctypes.data, b.ctypes.data, h, w, a.strides[0],
                                          u.ctypes.data, v.ctypes.data, u.strides[0], int(pyr_levels), int(win)))
        return u, v

    def enqueue_pair(self, frame_prev, frame_next, pyr_levels, win=3, out=None):
        """Asynchronous host-buffer call (lkflow_enqueue_pair_f64 / _u8 by dtype): returns the (U, V) arrays that
        wait() completes.  The inputs must stay alive and untouched until wait()."""
        u8 = isinstance(frame_prev, np.ndarray) and frame_prev.dtype == np.uint8
        a = self._host_u8(frame_prev) if u8 else self._host_in(frame_prev, False)
        b = self._host_u8(frame_next) if u8 else self._host_in(frame_next, False)
        h, w = a.shape
        u, v = out if out is not None else (np.empty((h, w)), np.empty((h, w)))
        if u8:
            rc = lib().lkflow_enqueue_pair_u8(self._h, a.ctypes.data, b.ctypes.data, h, w, a.strides[0],
                                              u.ctypes.data, v.ctypes.data, u.strides[0], int(pyr_levels), int(win))
        else:
            rc = lib().lkflow_enqueue_pair_f64(self._h, a.ctypes.data, b.ctypes.data, h, w, a.strides[0],
                                               u.ctypes.data, v.ctypes.data, u.strides[0], int(pyr_levels), int(win), 0)
        self._chk(rc)
        self._pending = (a, b, u, v)                  # keep the buffers alive until wait()
        return u, v

    def wait(self):
        self._chk(lib().lkflow_wait(self._h))
        self._pending = None

    def pin_host(self, on=True):
        """LKFLOW_OPT_PIN_HOST: page-lock the caller's (re-used) host buffers in place for asynchronous DMA."""
        self.set_option(LKFLOW_OPT_PIN_HOST, 1 if on else 0)

    @staticmethod
    def _host_u8(img):
        a = np.asarray(img)
        if a.dtype != np.uint8 or a.ndim != 2:
            raise TypeError("expected a 2-D uint8 image")
        if a.strides[1] != 1 or a.strides[0] < a.shape[1]:
            a = np.ascontiguousarray(a)
        return a

    def getLucasKanadeOpticalFlow(self, img1, img2, win=3):
        a, b = self._host_in(img1, False), self._host_in(img2, False)
        h, w = a.shape
        u, v = np.empty((h, w)), np.empty((h, w))
        self._chk(lib().lkflow_lk_f64(self._h, a.ctypes.data, b.ctypes.data, h, w, a.strides[0],
                                      u.ctypes.data, v.ctypes.data, u.strides[0], int(win)))
        return u, v

    @staticmethod
    def _host_in(img, inplace):
        if inplace:
            if not (isinstance(img, np.ndarray) and img.dtype == np.float64 and img.ndim == 2 and
                    img.strides[1] == 8 and img.flags.writeable):
                raise TypeError("mutate=True needs writable 2-D float64 arrays with unit inner stride")
            return img
        a = np.asarray(img, dtype=np.float64)
        if a.ndim != 2:
            raise ValueError("expected a 2-D image")
        if a.strides[1] != 8 or a.strides[0] < a.shape[1] * 8:
            a = np.ascontiguousarray(a)
        return a

    # ---- device (torch) batched forms -------------------------------------------------------
    def compute_batch(self, prev, nxt, levels, win=3, out=None, flags=0, stream=None):
        """prev, nxt: [P, H, W] float64 CUDA tensors (any plane/row strides, unit inner stride).
        Returns (U, V) [P, H, W].  Asynchronous on the current torch stream."""
        torch = _torch()
        _check_dev(prev, "prev"); _check_dev(nxt, "next")
        if prev.dim() == 2:
            prev, nxt = prev.unsqueeze(0), nxt.unsqueeze(0)
        P, H, W = prev.shape
        assert nxt.shape == prev.shape and nxt.stride(1) == prev.stride(1)
        if out is None:
            U = torch.empty((P, H, W), dtype=torch.float64, device=prev.device)
            V = torch.empty((P, H, W), dtype=torch.float64, device=prev.device)
        else:
            U, V = out
            _check_dev(U, "U"); _check_dev(V, "V")
            assert U.shape == (P, H, W) and V.shape == (P, H, W)
            assert U.stride() == V.stride()
        rc = lib().lkflow_compute_batch_f64_dev(
            self._h, prev.data_ptr(), nxt.data_ptr(), prev.stride(0) * 8, nxt.stride(0) * 8, P, H, W,
            prev.stride(1) * 8, U.data_ptr(), V.data_ptr(), U.stride(0) * 8, U.stride(1) * 8,
            int(levels), int(win), int(flags), _stream_ptr(stream))
        self._chk(rc)
        return U, V

    def compute_video(self, frames, levels, win=3, out=None, stream=None):
        """frames: [F, H, W] float64 CUDA tensor -> flow of the F-1 consecutive pairs."""
        return self.compute_batch(frames[:-1], frames[1:], levels, win, out=out, stream=stream)

    def compute_video_strict(self, frames, levels, win=3, stream=None):
        """The reference's video loop taken literally (ParallelVideoPyr.cpp:832-843): the level-0 warp
        overwrites both inputs of a pair and `prevDiff = diff` hands the WARPED frame to the next pair, so
        the pairs form a serial chain.  frames: [F, H, W] float64 CUDA tensor, overwritten exactly as the
        reference overwrites its frames.  Returns (U, V) [F-1, H, W].  One launch sequence per pair
        (no batching is possible); use compute_video() when pairs should be independent."""
        torch = _torch()
        _check_dev(frames, "frames")
        F, H, W = frames.shape
        U = torch.empty((F - 1, H, W), dtype=torch.float64, device=frames.device)
        V = torch.empty_like(U)
        for k in range(F - 1):
            self.compute_batch(frames[k:k + 1], frames[k + 1:k + 2], levels, win, out=(U[k:k + 1], V[k:k + 1]),
                               flags=LKFLOW_MUTATE_INPUTS, stream=stream)
        return U, V

    def lk_batch(self, img1, img2, win=3, stream=None):
        torch = _torch()
        _check_dev(img1, "img1"); _check_dev(img2, "img2")
        if img1.dim() == 2:
            img1, img2 = img1.unsqueeze(0), img2.unsqueeze(0)
        P, H, W = img1.shape
        assert img2.shape == img1.shape and img2.stride(1) == img1.stride(1)
        U = torch.empty((P, H, W), dtype=torch.float64, device=img1.device)
        V = torch.empty_like(U)
        rc = lib().lkflow_lk_batch_f64_dev(
            self._h, img1.data_ptr(), img2.data_ptr(), img1.stride(0) * 8, img2.stride(0) * 8, P, H, W,
            img1.stride(1) * 8, U.data_ptr(), V.data_ptr(), U.stride(0) * 8, U.stride(1) * 8, int(win),
            _stream_ptr(stream))
        self._chk(rc)
        return U, V

    def lk_up_batch(self, img1, img2, cu, cv, win=3, stream=None):
        """Single-level LK + "+= pyrUp(coarse flow)" (ParallelVideoPyr.cpp:634-637): img [P, H, W], cu / cv [P, H/2, W/2]."""
        torch = _torch()
        for t, n in ((img1, "img1"), (img2, "img2"), (cu, "cu"), (cv, "cv")):
            _check_dev(t, n)
        P, H, W = img1.shape
        assert img2.shape == img1.shape and img2.stride(1) == img1.stride(1)
        assert cu.shape == (P, H // 2, W // 2) and cv.shape == cu.shape and cu.stride() == cv.stride()
        U = torch.empty((P, H, W), dtype=torch.float64, device=img1.device)
        V = torch.empty_like(U)
        rc = lib().lkflow_lk_up_batch_f64_dev(
            self._h, img1.data_ptr(), img2.data_ptr(), img1.stride(0) * 8, img2.stride(0) * 8, P, H, W,
            img1.stride(1) * 8, cu.data_ptr(), cv.data_ptr(), cu.stride(0) * 8, cu.stride(1) * 8,
            U.data_ptr(), V.data_ptr(), U.stride(0) * 8, U.stride(1) * 8, int(win), _stream_ptr(stream))
        self._chk(rc)
        return U, V

    def pyrdown(self, src, stream=None):
        torch = _torch()
        _check_dev(src, "src")
        if src.dim() == 2:
            src = src.unsqueeze(0)
        N, H, W = src.shape
        dst = torch.empty((N, (H + 1) // 2, (W + 1) // 2), dtype=torch.float64, device=src.device)
        self._chk(lib().lkflow_pyrdown_f64_dev(self._h, src.data_ptr(), src.stride(0) * 8, N, H, W,
                                               src.stride(1) * 8, dst.data_ptr(), dst.stride(0) * 8,
                                               dst.stride(1) * 8, _stream_ptr(stream)))
        return dst

    def pyrdown2(self, src, stream=None):
        """Two pyramid levels in one pass: src [N, H, W] (H, W multiples of 4) -> ([N, H/2, W/2], [N, H/4, W/4])."""
        torch = _torch()
        _check_dev(src, "src")
        N, H, W = src.shape
        d1 = torch.empty((N, H // 2, W // 2), dtype=torch.float64, device=src.device)
        d2 = torch.empty((N, H // 4, W // 4), dtype=torch.float64, device=src.device)
        self._chk(lib().lkflow_pyrdown2_f64_dev(self._h, src.data_ptr(), src.stride(0) * 8, N, H, W, src.stride(1) * 8,
                                                d1.data_ptr(), d1.stride(0) * 8, d1.stride(1) * 8, d2.data_ptr(),
                                                d2.stride(0) * 8, d2.stride(1) * 8, _stream_ptr(stream)))
        return d1, d2

    def pyrup(self, src, stream=None):
        torch = _torch()
        _check_dev(src, "src")
        if src.dim() == 2:
            src = src.unsqueeze(0)
        N, H, W = src.shape
        dst = torch.empty((N, 2 * H, 2 * W), dtype=torch.float64, device=src.device)
        self._chk(lib().lkflow_pyrup_f64_dev(self._h, src.data_ptr(), src.stride(0) * 8, N, H, W,
                                             src.stride(1) * 8, dst.data_ptr(), dst.stride(0) * 8,
                                             dst.stride(1) * 8, _stream_ptr(stream)))
        return dst

    def warp(self, img1, img2, cu, cv, stream=None):
        """img1,img2: [P,H,W]; cu,cv: [P,H/2,W/2] coarse flow -> warped (img1, img2)."""
        torch = _torch()
        for n, t in (("img1", img1), ("img2", img2), ("cu", cu), ("cv", cv)):
            _check_dev(t, n)
        P, H, W = img1.shape
        assert img1.stride() == img2.stride() and cu.stride() == cv.stride()
        o1, o2 = torch.empty_like(img1, memory_format=torch.contiguous_format), None
        o2 = torch.empty_like(o1)
        self._chk(lib().lkflow_warp_f64_dev(
            self._h, img1.data_ptr(), img2.data_ptr(), img1.stride(0) * 8, P, H, W, img1.stride(1) * 8,
            cu.data_ptr(), cv.data_ptr(), cu.stride(0) * 8, cu.stride(1) * 8, o1.data_ptr(),
            o2.data_ptr(), o1.stride(0) * 8, o1.stride(1) * 8, _stream_ptr(stream)))
        return o1, o2

    def solve(self, S, plain=False, stream=None):
        """a6 as a stage: S [5, n] float64 CUDA tensor of window sums -> (u, v) [n]; plain=True: two IEEE divisions."""
        torch = _torch()
        _check_dev(S, "S")
        assert S.dim() == 2 and S.shape[0] == 5 and S.is_contiguous()
        n = S.shape[1]
        u = torch.empty(n, dtype=torch.float64, device=S.device); v = torch.empty_like(u)
        self._chk(lib().lkflow_solve_f64_dev(self._h, S.data_ptr(), n, u.data_ptr(), v.data_ptr(), int(bool(plain)),
                                             _stream_ptr(stream)))
        return u, v

    def u8_to_f64(self, src, stream=None):
        torch = _torch()
        assert src.is_cuda and src.dtype == torch.uint8 and src.is_contiguous()
        dst = torch.empty(src.shape, dtype=torch.float64, device=src.device)
        self._chk(lib().lkflow_u8_to_f64_dev(self._h, src.data_ptr(), dst.data_ptr(), src.numel(),
                                             _stream_ptr(stream)))
        return dst

    def resize_cubic(self, bgr_u8, rows=1080, cols=1080, stream=None):
        """[N,Hs,Ws,3] uint8 BGR frames -> [N,rows,cols,3] uint8: resize(.., Size(cols, rows), 0, 0, INTER_CUBIC)
        (ParallelVideoPyr.cpp:778; OpenCV 2.4 8-bit fixed-point arithmetic)."""
        torch = _torch()
        assert bgr_u8.is_cuda and bgr_u8.dtype == torch.uint8
        if bgr_u8.dim() == 3:
            bgr_u8 = bgr_u8.unsqueeze(0)
        assert bgr_u8.shape[-1] == 3 and bgr_u8.stride(-1) == 1 and bgr_u8.stride(2) == 3
        N, H, W, _ = bgr_u8.shape
        out = torch.empty((N, rows, cols, 3), dtype=torch.uint8, device=bgr_u8.device)
        self._chk(lib().lkflow_resize_cubic_bgr8_dev(self._h, bgr_u8.data_ptr(), bgr_u8.stride(0), N, H, W,
                                                     bgr_u8.stride(1), out.data_ptr(), out.stride(0), int(rows),
                                                     int(cols), out.stride(1), _stream_ptr(stream)))
        return out

    def blur_gray(self, bgr_u8, ksize=9, sigma=1.5, stream=None):
        """[N,H,W,3] uint8 BGR frames -> [N,H,W] uint8: GaussianBlur(ksize x ksize, sigma) on the colour frame,
        then BGR2GRAY (ParallelVideoPyr.cpp:786-787; OpenCV 2.4 8-bit fixed-point arithmetic)."""
        torch = _torch()
        assert bgr_u8.is_cuda and bgr_u8.dtype == torch.uint8
        if bgr_u8.dim() == 3:
            bgr_u8 = bgr_u8.unsqueeze(0)
        assert bgr_u8.shape[-1] == 3 and bgr_u8.stride(-1) == 1 and bgr_u8.stride(2) == 3
        N, H, W, _ = bgr_u8.shape
        out = torch.empty((N, H, W), dtype=torch.uint8, device=bgr_u8.device)
        self._chk(lib().lkflow_blur_gray_bgr8_dev(self._h, bgr_u8.data_ptr(), bgr_u8.stride(0), N, H, W,
                                                  bgr_u8.stride(1), int(ksize), float(sigma), out.data_ptr(),
                                                  out.stride(0), out.stride(1), _stream_ptr(stream)))
        return out

    def frame_diff(self, gray_u8, rate=0.3, stream=None):
        """[F,H,W] uint8 gray frames -> [F-1,H,W] uint8: frame[k+1] - rate*frame[k], saturated
        (ParallelVideoPyr.cpp:800, cv::addWeighted float arithmetic)."""
        torch = _torch()
        assert gray_u8.is_cuda and gray_u8.dtype == torch.uint8 and gray_u8.dim() == 3 and gray_u8.stride(-1) == 1
        F, H, W = gray_u8.shape
        assert F >= 2
        out = torch.empty((F - 1, H, W), dtype=torch.uint8, device=gray_u8.device)
        self._chk(lib().lkflow_frame_diff_u8_dev(self._h, gray_u8[1:].data_ptr(), gray_u8.data_ptr(),
                                                 gray_u8.stride(0), F - 1, H, W, gray_u8.stride(1), float(rate),
                                                 out.data_ptr(), out.stride(0), out.stride(1),
                                                 _stream_ptr(stream)))
        return out

    def preprocess_bgr(self, bgr_u8, ksize=9, sigma=1.5, rate=0.3, thresh=10, stream=None):
        """The video loop's whole per-frame chain after the resize (ParallelVideoPyr.cpp:786-813):
        [F,H,W,3] uint8 BGR frames -> [F-1,H,W] float64 edge images, the inputs of the hot call
        (consecutive edge images form the F-2 frame pairs)."""
        return self.diff_edge_map(self.blur_gray(bgr_u8, ksize, sigma, stream), rate, thresh, stream)

    def diff_edge_map(self, gray_u8, rate=0.3, thresh=10, stream=None):
        """frame_diff() + edge_map() fused (the difference image is never stored): [F,H,W] uint8 gray frames
        -> [F-1,H,W] float64 edge images."""
        torch = _torch()
        assert gray_u8.is_cuda and gray_u8.dtype == torch.uint8 and gray_u8.dim() == 3 and gray_u8.stride(-1) == 1
        F, H, W = gray_u8.shape
        assert F >= 2
        out = torch.empty((F - 1, H, W), dtype=torch.float64, device=gray_u8.device)
        self._chk(lib().lkflow_diff_edge_map_u8_dev(self._h, gray_u8[1:].data_ptr(), gray_u8.data_ptr(),
                                                    gray_u8.stride(0), F - 1, H, W, gray_u8.stride(1), float(rate),
                                                    int(thresh), out.data_ptr(), out.stride(0) * 8,
                                                    out.stride(1) * 8, _stream_ptr(stream)))
        return out

    def edge_map(self, diff_u8, thresh=10, stream=None):
        """[N,H,W] uint8 frame differences -> [N,H,W] float64 edge images (ParallelVideoPyr.cpp:803-813:
        threshold TOZERO, Sobel x + Sobel y, dilate x2, erode x2) = the hot call's inputs."""
        torch = _torch()
        assert diff_u8.is_cuda and diff_u8.dtype == torch.uint8 and diff_u8.stride(-1) == 1
        if diff_u8.dim() == 2:
            diff_u8 = diff_u8.unsqueeze(0)
        N, H, W = diff_u8.shape
        out = torch.empty((N, H, W), dtype=torch.float64, device=diff_u8.device)
        self._chk(lib().lkflow_edge_map_u8_dev(self._h, diff_u8.data_ptr(), diff_u8.stride(0), N, H, W,
                                               diff_u8.stride(1), int(thresh), out.data_ptr(),
                                               out.stride(0) * 8, out.stride(1) * 8, _stream_ptr(stream)))
        return out

    def flow_centroid(self, U, V, thresh=20.0, stream=None):
        """[P,H,W] flows -> [P,3] tensor (count, sum_x, sum_y) of pixels with |flow| >= thresh
        (ParallelVideoPyr.cpp:845-873)."""
        torch = _torch()
        _check_dev(U, "U"); _check_dev(V, "V")
        if U.dim() == 2:
            U, V = U.unsqueeze(0), V.unsqueeze(0)
        P, H, W = U.shape
        assert U.stride() == V.stride()
        out = torch.empty((P, 3), dtype=torch.float64, device=U.device)
        self._chk(lib().lkflow_flow_centroid_f64_dev(self._h, U.data_ptr(), V.data_ptr(),
                                                     U.stride(0) * 8, P, H, W, U.stride(1) * 8,
                                                     float(thresh), out.data_ptr(),
                                                     _stream_ptr(stream)))
        return out


# ---- the C-ABI frame-queue engine over a device list (lkflow_stream_*, include/lkflow.h) ---------
class StreamEngine:
    """lkflow_stream: one worker thread + context + stream per entry of `devices` (in ONE process); push() takes a
    [F, H, W] uint8 HOST array (numpy), centroids() waits and returns the [F-1, 3] rows, flow(p) one pair's U, V."""

    def __init__(self, devices, rows, cols, levels, win, keep_flow=False):
        arr = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        h = _vp()
        rc = lib().lkflow_stream_create(ctypes.byref(h), arr, len(devices), int(rows), int(cols), int(levels), int(win),
                                        LKFLOW_STREAM_KEEP_FLOW if keep_flow else 0)
        if rc:
            raise LKFlowError(rc)
        self._h, self.rows, self.cols, self._frames, self.n_pairs = h, rows, cols, None, 0

    def close(self):
        if self._h:
            lib().lkflow_stream_destroy(self._h)
            self._h = None
            self._frames = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _chk(self, rc):
        if rc:
            raise LKFlowError(rc, lib().lkflow_stream_last_error(self._h).decode())

    def set_budget(self, nbytes):
        self._chk(lib().lkflow_stream_set_budget(self._h, int(nbytes)))

    def push(self, frames_u8, thresh=20.0):
        import numpy as np
        f = frames_u8
        assert isinstance(f, np.ndarray) and f.dtype == np.uint8 and f.ndim == 3 and f.strides[2] == 1
        assert f.shape[1:] == (self.rows, self.cols)
        self._frames = f                                     # must stay alive until the push has completed
        self.n_pairs = f.shape[0] - 1
        self._chk(lib().lkflow_stream_push_u8(self._h, f.ctypes.data, f.shape[0], f.strides[0], f.strides[1], float(thresh)))

    def centroids(self):
        import numpy as np
        out = np.empty((self.n_pairs, 3), dtype=np.float64)
        self._chk(lib().lkflow_stream_centroids(self._h, out.ctypes.data, self.n_pairs))
        return out

    def flow(self, pair):
        import numpy as np
        U = np.empty((self.rows, self.cols)); V = np.empty_like(U)
        self._chk(lib().lkflow_stream_flow(self._h, int(pair), U.ctypes.data, V.ctypes.data, U.strides[0]))
        return U, V


def stream_block(n_frames, n_devices, k):
    """lkflow_stream_block: (first_frame, n_frames_local) of list entry k -- the C side of shard_pairs()."""
    a, b = _i(), _i()
    rc = lib().lkflow_stream_block(int(n_frames), int(n_devices), int(k), ctypes.byref(a), ctypes.byref(b))
    if rc:
        raise LKFlowError(rc)
    return a.value, b.value


# ---- frame-pair sharding of a video stream over ranks ---------------------------
def shard_pairs(n_frames, world_size, rank):
    """Contiguous block partition of the n_frames-1 consecutive pairs with a 1-frame overlap:
    returns (first_frame, n_frames_local); rank r computes pairs [first, first + n_local - 1).
    Blocks differ in size by at most one pair; trailing ranks may be empty (n_frames_local == 0)."""
    n_pairs = max(n_frames - 1, 0)
    base, rem = divmod(n_pairs, world_size)
    cnt = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return (first, cnt + 1) if cnt > 0 else (first, 0)
