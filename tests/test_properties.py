"""CPU suite: property tests of the oracle (hypothesis) + AddressSanitizer/UBSan build of the oracle.
These pin host-side invariants the GPU tests rely on."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@settings(max_examples=25, deadline=None)
@given(h=st.integers(1, 24), w=st.integers(1, 24), seed=st.integers(0, 2**31 - 1))
def test_pyramid_stage_shapes_and_linearity(oracle, h, w, seed):
    rng = np.random.default_rng(seed)
    a, b = rng.standard_normal((h, w)), rng.standard_normal((h, w))
    d = oracle.pyr_down(a)
    assert d.shape == ((h + 1) // 2, (w + 1) // 2)
    # linear filters: f(a + b) == f(a) + f(b) up to rounding
    assert np.allclose(oracle.pyr_down(a + b), d + oracle.pyr_down(b), atol=1e-12)
    u = oracle.pyr_up(a)
    assert u.shape == (2 * h, 2 * w)
    assert np.allclose(oracle.pyr_up(a + b), u + oracle.pyr_up(b), atol=1e-11)
    # both kernels have unit DC gain
    c = np.full((h, w), 1.25)
    assert np.allclose(oracle.pyr_down(c), 1.25, atol=1e-15) and np.allclose(oracle.pyr_up(c), 1.25, atol=1e-15)


@settings(max_examples=20, deadline=None)
@given(h=st.integers(3, 20), w=st.integers(3, 20), win=st.sampled_from([3, 5, 7]), seed=st.integers(0, 2**31 - 1))
def test_lk_zero_ring_no_nan_and_time_reversal(oracle, h, w, win, seed):
    rng = np.random.default_rng(seed)
    a, b = rng.random((h, w)), rng.random((h, w))
    u, v = oracle.lucas_kanade(a, b, win)
    assert np.isfinite(u).all() and np.isfinite(v).all()
    assert not u[0].any() and not u[-1].any() and not u[:, 0].any() and not u[:, -1].any()
    # swapping the frames flips ft and nothing else: flow changes sign exactly
    u2, v2 = oracle.lucas_kanade(b, a, win)
    assert np.array_equal(u2, -u) and np.array_equal(v2, -v)
    # identical frames: ft == 0 -> flow == 0 exactly (no NaN from 0/0: divide-by-zero rule)
    u0, v0 = oracle.lucas_kanade(a, a, win)
    assert not u0.any() and not v0.any()


@settings(max_examples=30, deadline=None)
@given(w=st.integers(1, 5000), h=st.integers(1, 5000))
def test_max_layer_is_the_largest_even_chain(oracle, w, h):
    L = oracle.max_layer(w, h)
    ww, hh = w, h
    for _ in range(L - 1):          # every level but the coarsest must have even dimensions
        assert ww % 2 == 0 and hh % 2 == 0
        ww //= 2; hh //= 2
    assert ww % 2 == 1 or hh % 2 == 1 or L >= 31


def test_oracle_under_address_and_ub_sanitizer(tmp_path):
    """Runs the golden-sized pyramid through an ASan+UBSan build of oracle/lk_oracle.c."""
    so = tmp_path / "liblk_oracle_asan.so"
    src = os.path.join(ROOT, "oracle", "lk_oracle.c")
    subprocess.check_call(["gcc", "-O1", "-g", "-fPIC", "-std=c11", "-ffp-contract=off", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-shared", "-o", str(so), src, "-lm"])
    code = f"""
import ctypes, numpy as np
L = ctypes.CDLL({str(so)!r})
dp = ctypes.POINTER(ctypes.c_double)
L.lko_coarse_to_fine.argtypes = [dp, dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, dp]
rng = np.random.default_rng(0)
for (h, w, lv, win) in [(448, 640, 7, 3), (40, 56, 3, 15), (8, 8, 4, 5), (2, 2, 2, 3), (1, 1, 1, 3)]:
    a, b = rng.random((h, w)), rng.random((h, w))
    u, v = np.empty((h, w)), np.empty((h, w))
    rc = L.lko_coarse_to_fine(a.ctypes.data_as(dp), b.ctypes.data_as(dp), h, w, lv, win, u.ctypes.data_as(dp), v.ctypes.data_as(dp))
    assert rc == 0 and np.isfinite(u).all()
bp = ctypes.POINTER(ctypes.c_ubyte)
for (h, w, k, sg) in [(1, 1, 9, 1.5), (3, 4, 9, 1.5), (9, 2, 15, 3.0), (40, 50, 3, 0.8)]:      # f2 chain, tiny / ragged shapes
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8); g = np.empty((h, w), np.uint8)
    L.lko_blur_gray_bgr8(img.ctypes.data_as(bp), h, w, k, ctypes.c_double(sg), g.ctypes.data_as(bp))
    d = np.empty((h, w), np.uint8); e = np.empty((h, w))
    L.lko_frame_diff_u8(g.ctypes.data_as(bp), img[..., 0].copy().ctypes.data_as(bp), ctypes.c_long(h * w), ctypes.c_double(0.3), d.ctypes.data_as(bp))
    L.lko_edge_map_u8(d.ctypes.data_as(bp), h, w, 10, e.ctypes.data_as(dp))
print("asan-ok")
"""
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "asan-ok" in r.stdout, r.stdout + r.stderr[-2000:]
