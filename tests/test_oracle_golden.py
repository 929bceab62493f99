"""CPU suite: pins the oracle against the reference's own golden vectors.

t1  U2Test/V2Test   single-level LK      (checker: parallelBasic.cpp:584-647)
t2  UPyrTest/VPyrTest 7-level pyramid    (checker: ParallelBasicPyr.cpp:773-836, generator SerialPyr.cpp:247-249)
t3  V2.txt          LK on the inputs mutated in place by the 7-level run (ctfLKof.cpp:246-254)
Tolerance = the reference's own bar, abs 1e-6 (ParallelBasicPyr.cpp:787).
"""
import numpy as np
import pytest

TOL = 1e-6  # ParallelBasicPyr.cpp:787 "double thresh = 0.000001"


def test_max_layer(oracle):
    # SURVEY.md 8(a) a1
    cases = {(640, 448): 7, (1080, 1080): 4, (1920, 1080): 4, (3840, 2160): 5, (7680, 4320): 6,
             (240, 160): 5, (3264, 2448): 5, (1, 1): 1, (3, 5): 1, (2, 2): 2}
    for (w, h), want in cases.items():
        assert oracle.max_layer(w, h) == want


def test_t1_single_level_golden(oracle, table_f64, goldens):
    u, v = oracle.lucas_kanade(*table_f64, 3)
    assert np.abs(u - goldens["U2Test"]).max() <= TOL
    assert np.abs(v - goldens["V2Test"]).max() <= TOL
    # invariants the goldens pin: zero border ring, no NaN/Inf
    assert np.isfinite(u).all() and np.isfinite(v).all()
    assert not u[0].any() and not u[-1].any() and not u[:, 0].any() and not u[:, -1].any()


def test_t2_t3_pyramid_golden_and_inplace_warp(oracle, table_f64, goldens):
    a, b = table_f64[0].copy(), table_f64[1].copy()
    u, v = oracle.coarse_to_fine(a, b, oracle.max_layer(640, 448), 3, mutate=True)
    assert np.abs(u - goldens["UPyrTest"]).max() <= TOL
    assert np.abs(v - goldens["VPyrTest"]).max() <= TOL
    # the reference's inputs were overwritten by the level-0 warp; V2.txt is LK on those
    assert not np.array_equal(a, table_f64[0])
    _, v2 = oracle.lucas_kanade(a, b, 3)
    assert np.abs(v2 - goldens["V2"]).max() <= TOL


def test_non_mutating_copy_gives_same_flow(oracle, table_f64, goldens):
    u, v = oracle.coarse_to_fine(table_f64[0], table_f64[1], 7, 3, mutate=False)
    assert np.abs(u - goldens["UPyrTest"]).max() <= TOL
    assert np.abs(v - goldens["VPyrTest"]).max() <= TOL


def test_level_and_window_validation(oracle, table_f64):
    with pytest.raises(ValueError):
        oracle.coarse_to_fine(*table_f64, 8, 3)      # > getMaxLayer: reference throws at :635
    with pytest.raises(ValueError):
        oracle.coarse_to_fine(*table_f64, 0, 3)
    with pytest.raises(ValueError):
        oracle.coarse_to_fine(*table_f64, 3, 4)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("levels,win", [(7, 3), (3, 15)])
def test_omp_baseline_is_bit_identical_to_oracle(oracle, table_f64, mode, levels, win):
    u, v = oracle.coarse_to_fine(*table_f64, levels, win)
    uo, vo = oracle.omp_coarse_to_fine(*table_f64, levels, win, mode)
    assert np.array_equal(u, uo) and np.array_equal(v, vo)


def test_zero_motion_gives_zero_flow(oracle):
    import synth
    a = synth.smooth_noise(64, 96, seed=7)
    u, v = oracle.coarse_to_fine(a, a, 3, 5)
    assert not u.any() and not v.any()


def test_translation_sign_and_magnitude(oracle):
    import synth
    a, b = synth.frame_pair(128, 192, seed=3, shift=(1.5, -0.75))
    # single level: I2(x) = I1(x - d) -> flow ~ +d (the reference's 2x2 derivative taps sit half a
    # pixel off-centre, so the estimate is biased by a few percent; the multi-level sum is NOT a
    # metric flow because the reference adds pyrUp(u) without the x2 rescale)
    u, v = oracle.coarse_to_fine(a, b, 1, 15)
    core = (slice(24, -24), slice(24, -24))
    assert abs(np.median(u[core]) - 1.5) < 0.25
    assert abs(np.median(v[core]) + 0.75) < 0.15


def test_window_generalisation_reduces_to_reference_at_3(oracle):
    rng = np.random.default_rng(0)
    m = rng.random((9, 11))
    s = oracle.sum_win(m, 3)
    want = np.zeros_like(m)
    for i in range(1, 8):
        for j in range(1, 10):
            want[i, j] = m[i - 1:i + 2, j - 1:j + 2].sum()
    assert np.allclose(s, want, rtol=0, atol=1e-14)
    # clipped window rule at win=5: inside the 1-px ring but window clipped -> m*win^2
    s5 = oracle.sum_win(m, 5)
    assert s5[1, 1] == m[1, 1] * 25 and s5[0, 3] == 0.0
    assert abs(s5[4, 5] - m[2:7, 3:8].sum()) < 1e-13


def test_pyr_shapes_and_constants(oracle):
    c = np.full((7, 10), 2.5)
    d = oracle.pyr_down(c)
    assert d.shape == (4, 5) and np.allclose(d, 2.5, atol=1e-15)
    up = oracle.pyr_up(c)
    assert up.shape == (14, 20) and np.allclose(up, 2.5, atol=1e-15)
    one = oracle.pyr_up(np.array([[3.0]]))
    assert one.shape == (2, 2) and np.allclose(one, 3.0)


def test_remap_identity_and_border(oracle):
    rng = np.random.default_rng(1)
    img = rng.random((6, 8))
    z = np.zeros((6, 8))
    assert np.array_equal(oracle.remap_linear(img, oracle.build_map(z, z, 1.0)), img)
    # a +2 px x-flow -> map1 = x + 1: last column samples outside -> border value 0 contributes
    two = np.full((6, 8), 2.0)
    w = oracle.remap_linear(img, oracle.build_map(two, z, 1.0))
    assert np.array_equal(w[:, :-1], img[:, 1:]) and not w[:, -1].any()
    # 1/32-px fixed point, round-half-even: 1/64 px rounds to 0
    tiny = np.full((6, 8), 2.0 / 64)
    assert np.array_equal(oracle.remap_linear(img, oracle.build_map(tiny, z, 1.0)), img)


def test_flow_centroid(oracle):
    u = np.zeros((5, 7)); v = np.zeros((5, 7))
    u[2, 3] = 25.0; v[4, 6] = -30.0; u[0, 0] = 19.999
    c, sx, sy = oracle.flow_centroid(u, v, 20.0)
    assert (c, sx, sy) == (2, 9.0, 6.0)


def test_edge_map_oracle_matches_direct_5x5_formulation(oracle):
    """ParallelVideoPyr.cpp:803-813: two 3x3 dilations (erosions) with outside pixels ignored == one 5x5
    max (min); Sobel x + Sobel y with reflect-101; threshold TOZERO.  Brute-force NumPy cross-check."""
    rng = np.random.default_rng(0)
    d = rng.integers(0, 256, (20, 30)).astype(np.uint8)
    got = oracle.edge_map_u8(d, 10)
    t = np.where(d > 10, d, 0).astype(np.float64)
    H, W = t.shape

    def r(i, n):
        while i < 0 or i >= n:
            i = -i if i < 0 else 2 * n - 2 - i
        return i
    e = np.zeros_like(t)
    for y in range(H):
        for x in range(W):
            ym, yp, xm, xp = r(y - 1, H), r(y + 1, H), r(x - 1, W), r(x + 1, W)
            sx = (t[ym, xp] - t[ym, xm]) + 2 * (t[y, xp] - t[y, xm]) + (t[yp, xp] - t[yp, xm])
            sy = (t[yp, xm] - t[ym, xm]) + 2 * (t[yp, x] - t[ym, x]) + (t[yp, xp] - t[ym, xp])
            e[y, x] = sx + sy

    def mm(a, f):
        return np.array([[f(a[max(0, y - 2):y + 3, max(0, x - 2):x + 3]) for x in range(W)] for y in range(H)])
    assert np.array_equal(got, mm(mm(e, np.max), np.min))
    assert not oracle.edge_map_u8(np.full((9, 9), 7, np.uint8), 10).any()      # below threshold -> flat -> 0


# ------------------------------------------------------------------ f2 pre-processing (PARITY UNPINNED)
# The reference holds no fixture for ParallelVideoPyr.cpp:786-800 and OpenCV is absent: these tests pin the
# C oracle to an independent NumPy/SciPy restatement of the OpenCV 2.4 fixed-point algorithms it documents.
def test_gaussian_kernel_q8(oracle):
    k = oracle.gaussian_kernel_q8(9, 1.5)
    assert k.tolist() == [2, 9, 28, 55, 68, 55, 28, 9, 2] and k.sum() == 256
    k5 = oracle.gaussian_kernel_q8(5, 1.0)
    assert k5.tolist() == k5[::-1].tolist() and abs(int(k5.sum()) - 256) <= 2


@pytest.mark.parametrize("shape", [(40, 50), (3, 4), (1, 1), (9, 2)])
def test_blur_gray_oracle_matches_numpy(oracle, shape):
    from scipy.ndimage import correlate1d
    rng = np.random.default_rng(shape[0] * 13 + shape[1])
    img = rng.integers(0, 256, shape + (3,)).astype(np.uint8)
    k = oracle.gaussian_kernel_q8(9, 1.5)

    def r101(i, n):
        if n == 1:
            return 0
        while i < 0 or i >= n:
            i = -i if i < 0 else 2 * n - 2 - i
        return i

    def corr(a, axis):      # reflect-101 correlation with multi-reflection for images shorter than the radius
        n = a.shape[axis]
        idx = np.array([[r101(i + j, n) for j in range(-4, 5)] for i in range(n)])
        g = np.take(a, idx, axis=axis)                 # axis -> (n, 9)
        return np.tensordot(g, k, axes=([axis + 1], [0]))
    c = corr(corr(img.astype(np.int64), 1), 0)
    if min(shape) > 4:                                 # scipy's 'mirror' == reflect-101 when one reflection suffices
        assert np.array_equal(c, correlate1d(correlate1d(img.astype(np.int64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror"))
    q, rem = c >> 16, c & 0xFFFF
    q = np.clip(q + ((rem > 0x8000) | ((rem == 0x8000) & ((q & 1) == 1))), 0, 255)
    gray = (q[..., 0] * 1868 + q[..., 1] * 9617 + q[..., 2] * 4899 + 8192) >> 14
    assert np.array_equal(oracle.blur_gray_bgr8(img), gray.astype(np.uint8))


def test_blur_gray_constant_frame(oracle):
    img = np.empty((12, 17, 3), np.uint8)
    img[..., 0], img[..., 1], img[..., 2] = 10, 200, 90           # B, G, R
    want = (10 * 1868 + 200 * 9617 + 90 * 4899 + 8192) >> 14
    assert (oracle.blur_gray_bgr8(img) == want).all()            # kernel sums to 256: a flat frame is unchanged


def test_frame_diff_oracle_matches_float32_numpy(oracle):
    a, b = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8))   # every (cur, prev) pair
    t = a.astype(np.float32) * np.float32(1) + b.astype(np.float32) * np.float32(-0.3)
    want = np.clip(np.rint(t), 0, 255).astype(np.uint8)
    got = oracle.frame_diff_u8(a, b, 0.3)
    assert np.array_equal(got, want)
    assert got[255, 0] == 0 and got[0, 255] == 255               # saturates at 0; cur alone passes through


# ---- f2: resize(.., Size(1080, 1080), 0, 0, INTER_CUBIC) on the 8UC3 frame (ParallelVideoPyr.cpp:778) -- parity unpinned ----
def _np_cubic_tab(ssize, dsize):
    """Independent NumPy form of OpenCV 2.4's cubic tables (imgwarp.cpp cv::resize + interpolateCubic), float32 per operation."""
    f32 = np.float32
    scale = 1.0 / (float(dsize) / ssize)
    f = ((np.arange(dsize, dtype=np.float64) + 0.5) * scale - 0.5).astype(f32)
    s = np.floor(f).astype(np.int32)
    x = f - s.astype(f32)
    A = f32(-0.75)
    x1, xm = x + f32(1), f32(1) - x
    c0 = ((A * x1 - f32(5) * A) * x1 + f32(8) * A) * x1 - f32(4) * A
    c1 = ((A + f32(2)) * x - (A + f32(3))) * x * x + f32(1)
    c2 = ((A + f32(2)) * xm - (A + f32(3))) * xm * xm + f32(1)
    c3 = f32(1) - c0 - c1 - c2
    c = np.stack([c0, c1, c2, c3], axis=1)
    assert c.dtype == np.float32
    return s, np.clip(np.rint(c * f32(2048)), -32768, 32767).astype(np.int32)


def _np_resize_cubic(src, drows, dcols):
    srows, scols, _ = src.shape
    xo, al = _np_cubic_tab(scols, dcols)
    yo, be = _np_cubic_tab(srows, drows)
    xi = np.clip(xo[:, None] + np.arange(-1, 3)[None, :], 0, scols - 1)            # [dcols, 4]
    H = (src.astype(np.int32)[:, xi, :] * al[None, :, :, None]).sum(axis=2)         # [srows, dcols, 3] int32
    H = H.reshape(srows, dcols * 3)
    yi = np.clip(yo[:, None] + np.arange(-1, 3)[None, :], 0, srows - 1)            # [drows, 4]
    S = H[yi]                                                                       # [drows, 4, width]
    f32 = np.float32
    b = be.astype(f32) * (f32(1) / (f32(2048) * f32(2048)))
    Sf = S.astype(f32)
    acc = Sf[:, 0] * b[:, 0:1]
    for k in (1, 2, 3):
        acc = acc + Sf[:, k] * b[:, k:k + 1]
    vf = np.rint(acc).astype(np.int64)
    vi = ((S.astype(np.int64) * be[:, :, None]).sum(axis=1) + (1 << 21)) >> 22
    width = dcols * 3
    v = np.where(np.arange(width)[None, :] < (width & ~7), vf, vi)
    return np.clip(v, 0, 255).astype(np.uint8).reshape(drows, dcols, 3)


@pytest.mark.parametrize("src_shape,dst_shape", [((48, 64), (108, 108)), ((72, 128), (54, 54)), ((30, 40), (37, 101)),
                                                  ((5, 7), (40, 33)), ((1, 1), (9, 9)), ((2, 3), (3, 2))])
def test_resize_cubic_matches_independent_numpy_form(oracle, src_shape, dst_shape):
    rng = np.random.default_rng(src_shape[0] * 1000 + dst_shape[1])
    src = rng.integers(0, 256, size=src_shape + (3,), dtype=np.uint8)
    got = oracle.resize_cubic_bgr8(src, *dst_shape)
    assert np.array_equal(got, _np_resize_cubic(src, *dst_shape))


def test_resize_cubic_known_answers(oracle):
    """What any reading of the algorithm must give: a same-size resize is the identity (weights 0, 2048, 0, 0), a flat
    image stays flat, the 1/2-pixel phase of a 2x up-sampling has the textbook weights (-3/32, 19/32, 19/32, -3/32) * ... in
    Q11 = the cubic kernel with A = -0.75 at x = 0.25 / 0.75."""
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, size=(31, 45, 3), dtype=np.uint8)
    assert np.array_equal(oracle.resize_cubic_bgr8(src, 31, 45), src)
    flat = np.full((20, 30, 3), 200, dtype=np.uint8)
    assert (oracle.resize_cubic_bgr8(flat, 47, 71) == 200).all()
    ofs, coef = oracle.resize_cubic_tables(100, 200)
    # d = 2k: f = k - 0.25 -> s = k - 1, x = 0.75; d = 2k + 1: f = k + 0.25 -> s = k, x = 0.25
    w = lambda x: np.array([((-0.75 * (x + 1) + 3.75) * (x + 1) - 6) * (x + 1) + 3, (1.25 * x - 2.25) * x * x + 1,
                            (1.25 * (1 - x) - 2.25) * (1 - x) ** 2 + 1, 0.0])
    for d, (s, x) in {10: (4, 0.75), 11: (5, 0.25)}.items():
        ww = w(x); ww[3] = 1 - ww[:3].sum()
        assert ofs[d] == s and np.array_equal(coef[d], np.rint(ww * 2048).astype(np.int16))
    assert (coef.sum(axis=1) == 2048).all()
