"""GPU suite (-m gpu): the HIP path, called through the C ABI, against the oracle / goldens.

Bars
  * byte-for-byte (np.array_equal) for the stages whose expression order the kernels mirror exactly:
    pyrDown, pyrUp, map build + fixed-point remap, uint8 conversion, centroid sums.
  * abs 1e-6 -- the reference's own bar (ParallelBasicPyr.cpp:787) -- for everything that contains
    the win x win window sum: the kernels add the same terms in a different (separable) order, so
    results differ from the oracle at rounding level (typically < 1e-11 on textured input).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-6        # ParallelBasicPyr.cpp:787
TIGHT = 1e-9      # expected agreement on well-conditioned synthetic input


@pytest.fixture(scope="module")
def torch():
    import torch
    return torch


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


# ---------------------------------------------------------------------------------------- stages
@pytest.mark.parametrize("shape", [(448, 640), (7, 10), (135, 240), (5, 3), (2, 2), (1, 1), (1, 9), (9, 1)])
def test_pyrdown_bit_exact(gpu_ctx, oracle, torch, shape):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    a = rng.random((3,) + shape)
    got = host(gpu_ctx.pyrdown(dev(torch, a)))
    for k in range(3):
        assert np.array_equal(got[k], oracle.pyr_down(a[k]))


@pytest.mark.parametrize("shape", [(224, 320), (7, 10), (1, 1), (1, 5), (5, 1), (2, 2), (3, 2)])
def test_pyrup_bit_exact(gpu_ctx, oracle, torch, shape):
    rng = np.random.default_rng(shape[0] * 77 + shape[1])
    a = rng.standard_normal((2,) + shape) * 10
    got = host(gpu_ctx.pyrup(dev(torch, a)))
    for k in range(2):
        assert np.array_equal(got[k], oracle.pyr_up(a[k]))


@pytest.mark.parametrize("shape,scale", [((64, 96), 3.0), ((14, 20), 40.0), ((2, 2), 1.0), ((448, 640), 100.0)])
def test_warp_bit_exact(gpu_ctx, oracle, torch, shape, scale):
    h, w = shape
    rng = np.random.default_rng(h + w)
    i1, i2 = rng.random((2, h, w)), rng.random((2, h, w))
    cu = rng.standard_normal((2, h // 2, w // 2)) * scale
    cv = rng.standard_normal((2, h // 2, w // 2)) * scale
    if h >= 14:
        cu[0, 0, 0] = 1e12; cv[0, 1, 1] = -1e12; cu[1, 2, 2] = 70000.0   # saturation / overflow paths
    o1, o2 = gpu_ctx.warp(dev(torch, i1), dev(torch, i2), dev(torch, cu), dev(torch, cv))
    o1, o2 = host(o1), host(o2)
    for k in range(2):
        upu, upv = oracle.pyr_up(cu[k]), oracle.pyr_up(cv[k])
        assert np.array_equal(o1[k], oracle.remap_linear(i1[k], oracle.build_map(upu, upv, 1.0)))
        assert np.array_equal(o2[k], oracle.remap_linear(i2[k], oracle.build_map(upu, upv, -1.0)))


def test_u8_conversion_bit_exact(gpu_ctx, oracle, torch, table_u8):
    got = host(gpu_ctx.u8_to_f64(dev(torch, table_u8[0])))
    assert np.array_equal(got, oracle.u8_to_f64(table_u8[0]))


# ------------------------------------------------------------------------------ single-level LK
@pytest.mark.parametrize("win", [3, 5, 7, 15, 21, 31])
@pytest.mark.parametrize("shape", [(96, 300), (40, 60)])
def test_lk_single_level_vs_oracle(gpu_ctx, oracle, torch, win, shape):
    import synth
    a, b = synth.frame_pair(*shape, seed=win)
    u, v = gpu_ctx.lk_batch(dev(torch, a), dev(torch, b), win)
    uo, vo = oracle.lucas_kanade(a, b, win)
    assert np.abs(host(u)[0] - uo).max() <= TIGHT
    assert np.abs(host(v)[0] - vo).max() <= TIGHT
    # structural invariants: outermost ring is exactly zero
    uu = host(u)[0]
    assert not uu[0].any() and not uu[-1].any() and not uu[:, 0].any() and not uu[:, -1].any()


@pytest.mark.parametrize("shape", [(3, 3), (2, 5), (1, 7), (7, 1), (4, 300), (300, 4), (17, 513)])
def test_lk_tiny_and_ragged_shapes(gpu_ctx, oracle, torch, shape):
    rng = np.random.default_rng(5)
    a, b = rng.random(shape), rng.random(shape)
    for win in (3, 9):
        u, v = gpu_ctx.lk_batch(dev(torch, a), dev(torch, b), win)
        uo, vo = oracle.lucas_kanade(a, b, win)
        assert np.allclose(host(u)[0], uo, rtol=1e-9, atol=TIGHT)
        assert np.allclose(host(v)[0], vo, rtol=1e-9, atol=TIGHT)


def test_t1_single_level_golden(gpu_ctx, table_f64, goldens):
    # parallelBasic.cpp:569-647 : getLucasKanadeOpticalFlow vs U2Test/V2Test
    u, v = gpu_ctx.getLucasKanadeOpticalFlow(*table_f64)
    assert np.abs(u - goldens["U2Test"]).max() <= TOL
    assert np.abs(v - goldens["V2Test"]).max() <= TOL
    assert np.isfinite(u).all() and np.isfinite(v).all()


def test_marching_kernel_matches_per_pixel_kernel(gpu_ctx, lkflow, torch, table_f64):
    a, b = dev(torch, table_f64[0]), dev(torch, table_f64[1])
    for win in (3, 15):
        u, v = gpu_ctx.lk_batch(a, b, win)
        gpu_ctx.set_option(lkflow.LKFLOW_OPT_IMPL, 1)
        try:
            un, vn = gpu_ctx.lk_batch(a, b, win)
        finally:
            gpu_ctx.set_option(lkflow.LKFLOW_OPT_IMPL, 0)
        assert (u - un).abs().max().item() <= TOL and (v - vn).abs().max().item() <= TOL


# ---------------------------------------------------------------------------------- full pyramid
def test_t2_t3_pyramid_golden_with_inplace_warp(gpu_ctx, lkflow, table_f64, goldens):
    # ParallelBasicPyr.cpp:717-836 (config 0a): 640x448 crop, getMaxLayer = 7 levels, 3x3
    a, b = table_f64[0].copy(), table_f64[1].copy()
    levels = lkflow.getMaxLayer(a)
    assert levels == 7
    u, v = gpu_ctx.coarseToFineEstimation(a, b, levels)       # mutate=True like the reference
    assert np.abs(u - goldens["UPyrTest"]).max() <= TOL
    assert np.abs(v - goldens["VPyrTest"]).max() <= TOL
    assert not np.array_equal(a, table_f64[0])                # inputs were overwritten (:696,:701)
    _, v2 = gpu_ctx.getLucasKanadeOpticalFlow(a, b)           # t3: V2.txt
    assert np.abs(v2 - goldens["V2"]).max() <= TOL


def test_mutated_inputs_equal_oracle_warp(gpu_ctx, oracle, table_f64):
    a, b = table_f64[0].copy(), table_f64[1].copy()
    ao, bo = table_f64[0].copy(), table_f64[1].copy()
    gpu_ctx.coarseToFineEstimation(a, b, 4, win=5)
    oracle.coarse_to_fine(ao, bo, 4, 5, mutate=True)
    assert np.abs(a - ao).max() <= TOL and np.abs(b - bo).max() <= TOL


def test_default_does_not_mutate(gpu_ctx, table_f64, goldens):
    a, b = table_f64[0].copy(), table_f64[1].copy()
    u, v = gpu_ctx.computeFlow(a, b, 7, 3)
    assert np.array_equal(a, table_f64[0]) and np.array_equal(b, table_f64[1])
    assert np.abs(u - goldens["UPyrTest"]).max() <= TOL


# BASELINE.json config shapes at oracle-friendly sizes (same level count / window)
@pytest.mark.parametrize("shape,levels,win", [
    ((272, 480), 3, 15),     # config 1 (1920x1080 L3 w15) / 4
    ((272, 480), 4, 3),      # reference's shipped loop: 4 levels, 3x3
    ((288, 512), 5, 21),     # config 2 (3840x2160 L5 w21)
    ((160, 240), 3, 15),     # config 3 (pacman 240x160)
    ((288, 512), 5, 31),     # config 4 (7680x4320 L5 w31)
    ((612, 816), 3, 15),     # config 0b (hand 3264x2448 L3 w15) / 4
])
def test_pyramid_vs_oracle_config_shapes(gpu_ctx, oracle, shape, levels, win):
    import synth
    a, b = synth.frame_pair(*shape, seed=levels * 100 + win)
    u, v = gpu_ctx.computeFlow(a, b, levels, win)
    uo, vo = oracle.coarse_to_fine(a, b, levels, win)
    assert np.abs(u - uo).max() <= TOL and np.abs(v - vo).max() <= TOL
    # report how tight it really is
    print(f"{shape} L{levels} w{win}: max|du|={np.abs(u - uo).max():.3e} max|dv|={np.abs(v - vo).max():.3e}")


def test_full_size_1080p_vs_oracle(gpu_ctx, oracle):
    # config 1 at FULL size: 1920x1080, 3 levels, 15x15 (oracle takes a few seconds)
    import synth
    a, b = synth.frame_pair(1080, 1920, seed=1234)
    u, v = gpu_ctx.computeFlow(a, b, 3, 15)
    uo, vo = oracle.coarse_to_fine(a, b, 3, 15)
    assert np.abs(u - uo).max() <= TOL and np.abs(v - vo).max() <= TOL


def test_full_size_4k_march_equals_per_pixel(gpu_ctx, lkflow, torch):
    # config 2 at FULL size (3840x2160, 5 levels, 21x21): the oracle would need minutes, so the
    # size-independent check is tiled-kernel == per-pixel-kernel (reference summation order).
    import synth
    prev, nxt = synth.frame_pairs_torch(1, 2160, 3840, "cuda", seed=7)
    u, v = gpu_ctx.compute_batch(prev, nxt, 5, 21)
    gpu_ctx.set_option(lkflow.LKFLOW_OPT_IMPL, 1)
    try:
        un, vn = gpu_ctx.compute_batch(prev, nxt, 5, 21)
    finally:
        gpu_ctx.set_option(lkflow.LKFLOW_OPT_IMPL, 0)
    torch.cuda.synchronize()
    assert (u - un).abs().max().item() <= TOL and (v - vn).abs().max().item() <= TOL


def test_zero_motion_is_exactly_zero(gpu_ctx, torch):
    import synth
    a = dev(torch, synth.smooth_noise(270, 480, seed=2))
    u, v = gpu_ctx.compute_batch(a.unsqueeze(0), a.unsqueeze(0).clone(), 2, 15)
    assert not u.any().item() and not v.any().item()


# ------------------------------------------------------------------- batching, strides, sharding
def test_batch_equals_single_bitwise_and_video_api(gpu_ctx, oracle, torch):
    import synth
    frames = synth.video(6, 160, 240, seed=11)                  # pacman-shaped stream
    d = dev(torch, frames)
    U, V = gpu_ctx.compute_video(d, 3, 15)
    assert U.shape == (5, 160, 240)
    for p in range(5):
        u1, v1 = gpu_ctx.compute_batch(d[p:p + 1], d[p + 1:p + 2], 3, 15)
        assert torch.equal(u1[0], U[p]) and torch.equal(v1[0], V[p])
    uo, vo = oracle.coarse_to_fine(frames[2], frames[3], 3, 15)
    assert np.abs(host(U[2]) - uo).max() <= TOL and np.abs(host(V[2]) - vo).max() <= TOL
    assert np.array_equal(host(d), frames)                      # inputs untouched


def test_strided_inputs_and_outputs(gpu_ctx, oracle, torch):
    import synth
    a, b = synth.frame_pair(64, 96, seed=4)
    big = torch.zeros((2, 70, 128), dtype=torch.float64, device="cuda")
    big[0, :64, 5:101] = dev(torch, a)
    big[1, :64, 5:101] = dev(torch, b)
    pa, pb = big[0:1, :64, 5:101], big[1:2, :64, 5:101]         # row stride 128, offset 5
    outb = torch.full((2, 64, 200), -7.0, dtype=torch.float64, device="cuda")
    U, V = outb[0:1, :, 3:99], outb[1:2, :, 3:99]
    # U and V must share strides; they do (same parent)
    gpu_ctx.compute_batch(pa, pb, 2, 5, out=(U, V))
    uo, vo = oracle.coarse_to_fine(a, b, 2, 5)
    assert np.abs(host(U)[0] - uo).max() <= TIGHT and np.abs(host(V)[0] - vo).max() <= TIGHT
    assert (outb[:, :, :3] == -7.0).all().item() and (outb[:, :, 99:] == -7.0).all().item()


def test_flow_centroid_matches_oracle(gpu_ctx, oracle, torch):
    rng = np.random.default_rng(9)
    u = rng.standard_normal((3, 90, 130)) * 15
    v = rng.standard_normal((3, 90, 130)) * 15
    out = host(gpu_ctx.flow_centroid(dev(torch, u), dev(torch, v), 20.0))
    for p in range(3):
        c, sx, sy = oracle.flow_centroid(u[p], v[p], 20.0)
        assert tuple(out[p]) == (float(c), sx, sy)


@pytest.mark.parametrize("thresh", [20.0, 3.3, 1e-3, 0.0, -1.0, float("inf"), float("nan")])
def test_flow_centroid_threshold_boundary(gpu_ctx, oracle, torch, thresh):
    """The kernel compares u*u+v*v with a pre-squared threshold instead of taking sqrt per pixel: the decisions
    must be those of `sqrt(u*u+v*v) < thresh` (ParallelVideoPyr.cpp:856-858) exactly, also within a few ulps of
    the circle |flow| == thresh, and for degenerate thresholds / non-finite flow."""
    rng = np.random.default_rng(3)
    r = thresh if np.isfinite(thresh) and thresh > 0 else 1.0
    th = rng.uniform(0, 2 * np.pi, (2, 64, 200))
    eps = rng.integers(-3, 4, th.shape) * np.spacing(r)            # radii within +-3 ulp of the threshold
    u, v = (r + eps) * np.cos(th), (r + eps) * np.sin(th)
    u[0, 0, :4] = (r, -r, 0.0, 0.0); v[0, 0, :4] = (0.0, 0.0, r, -r)  # exactly on the circle
    u[1, 1, :3] = (np.nan, np.inf, 0.0); v[1, 1, :3] = (0.0, 1.0, -np.inf)
    out = host(gpu_ctx.flow_centroid(dev(torch, u), dev(torch, v), thresh))
    for p in range(2):
        c, sx, sy = oracle.flow_centroid(u[p], v[p], thresh)
        assert tuple(out[p]) == (float(c), sx, sy)


@pytest.mark.parametrize("n,offset", [(1, 0), (2, 0), (7, 1), (4096 + 5, 0), (100003, 1), (640 * 448, 2)])
def test_u8_conversion_ragged(gpu_ctx, oracle, torch, n, offset):
    """Odd lengths / odd source addresses take the scalar head-tail path of the vectorised kernel."""
    rng = np.random.default_rng(n)
    buf = torch.from_numpy(rng.integers(0, 256, n + offset).astype(np.uint8)).cuda()
    src = buf[offset:]
    assert np.array_equal(host(gpu_ctx.u8_to_f64(src)), oracle.u8_to_f64(host(src)))


# ------------------------------------------------------------------------------- error behaviour
def test_error_codes(gpu_ctx, lkflow, torch):
    a = torch.zeros((1, 64, 96), dtype=torch.float64, device="cuda")
    with pytest.raises(lkflow.LKFlowError) as e:
        gpu_ctx.compute_batch(a, a.clone(), 7, 3)               # > getMaxLayer(96,64) = 6
    assert e.value.status == lkflow.LKFLOW_E_LEVELS
    with pytest.raises(lkflow.LKFlowError) as e:
        gpu_ctx.compute_batch(a, a.clone(), 0, 3)
    assert e.value.status == lkflow.LKFLOW_E_LEVELS
    for win in (4, 1, 33):
        with pytest.raises(lkflow.LKFlowError) as e:
            gpu_ctx.compute_batch(a, a.clone(), 2, win)
        assert e.value.status == lkflow.LKFLOW_E_WIN
    two = torch.zeros((2, 64, 96), dtype=torch.float64, device="cuda")
    with pytest.raises(lkflow.LKFlowError) as e:                # MUTATE only for a single pair
        gpu_ctx.compute_batch(two, two.clone(), 2, 3, flags=lkflow.LKFLOW_MUTATE_INPUTS)
    assert e.value.status == lkflow.LKFLOW_E_ARG
    L = lkflow.lib()
    assert L.lkflow_compute_batch_f64_dev(gpu_ctx._h, a.data_ptr(), a.data_ptr(), 0, 0, 1, 64, 96, 8 * 95,
                                          a.data_ptr(), a.data_ptr(), 0, 8 * 96, 1, 3, 0,
                                          None) == lkflow.LKFLOW_E_ARG   # step < cols*8


def test_stream_leg_u8_frames_to_flow_and_centroid(gpu_ctx, oracle, torch):
    """The per-rank GPU leg of the sharded video path (sharding.stream_flow): uint8 frames in,
    convertTo(1/255) + flow of consecutive pairs + fused centroid rows out."""
    import sharding
    import synth
    v = synth.video(4, 160, 240, seed=21)
    fr = np.clip(np.rint(v * 255), 0, 255).astype(np.uint8)
    U, V, rows = sharding.stream_flow(dev(torch, fr), gpu_ctx, 3, 15, thresh=0.5)
    assert U.shape == (3, 160, 240) and rows.shape == (3, 3)
    for p in range(3):
        a, b = oracle.u8_to_f64(fr[p]), oracle.u8_to_f64(fr[p + 1])
        uo, vo = oracle.coarse_to_fine(a, b, 3, 15)
        assert np.abs(host(U[p]) - uo).max() <= TOL and np.abs(host(V[p]) - vo).max() <= TOL
        c, sx, sy = oracle.flow_centroid(host(U[p]), host(V[p]), 0.5)
        assert tuple(host(rows[p])) == (float(c), sx, sy)


def test_fuzz_random_shapes_windows_levels(gpu_ctx, lkflow, oracle):
    """Seeded fuzz over ragged sizes (odd widths/heights, images narrower than a window, images
    smaller than one tile, several pyramid depths): pyramid and single-level LK vs the oracle."""
    rng = np.random.default_rng(20261004)
    wins = [3, 5, 7, 9, 11, 13, 15, 17, 21, 25, 31]
    worst = 0.0
    for case in range(48):
        win = int(rng.choice(wins))
        if case % 3 == 0:                                     # single level, arbitrary (odd) sizes
            h, w = int(rng.integers(1, 70)), int(rng.integers(1, 600))
            a, b = rng.random((h, w)), rng.random((h, w))
            u, v = gpu_ctx.getLucasKanadeOpticalFlow(a, b, win)
            uo, vo = oracle.lucas_kanade(a, b, win)
        else:                                                 # pyramids: sizes = odd * 2^(L-1)
            levels = int(rng.integers(2, 6))
            h = int(rng.integers(1, 12)) * 2 ** (levels - 1)
            w = int(rng.integers(1, 40)) * 2 ** (levels - 1)
            assert lkflow.getMaxLayer(w, h) >= levels
            a, b = rng.random((h, w)), rng.random((h, w))
            u, v = gpu_ctx.computeFlow(a, b, levels, win)
            uo, vo = oracle.coarse_to_fine(a, b, levels, win)
        # white-noise inputs with tiny windows are ill-conditioned: compare relative to the flow scale
        scale = max(1.0, np.abs(uo).max(), np.abs(vo).max())
        err = max(np.abs(u - uo).max(), np.abs(v - vo).max()) / scale
        worst = max(worst, err)
        assert err <= 1e-6, (case, h, w, win, err)
    print(f"fuzz worst relative error {worst:.3e}")


def test_config0b_hand_images_vs_oracle(gpu_ctx, oracle):
    """BASELINE config 0b: real camera frames (hand1/hand2.jpg, centred 816x612 crop), 3 levels,
    15x15 window.  The reference never reads these files and holds no golden for them, so the check
    is HIP path vs the (golden-pinned) oracle at the reference's own bar."""
    import os
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    a = oracle.u8_to_f64(np.fromfile(os.path.join(g, "hand1_816x612.u8"), dtype=np.uint8).reshape(612, 816))
    b = oracle.u8_to_f64(np.fromfile(os.path.join(g, "hand2_816x612.u8"), dtype=np.uint8).reshape(612, 816))
    u, v = gpu_ctx.computeFlow(a, b, 3, 15)
    uo, vo = oracle.coarse_to_fine(a, b, 3, 15)
    scale = max(1.0, np.abs(uo).max(), np.abs(vo).max())
    assert np.abs(u - uo).max() <= TOL * scale and np.abs(v - vo).max() <= TOL * scale
    print(f"hand crop: max|du|={np.abs(u - uo).max():.3e} (flow range {np.abs(uo).max():.1f} px)")


def test_full_size_8k_w31_march_equals_per_pixel(gpu_ctx, lkflow, torch):
    """BASELINE config 4 shape at FULL size (7680x4320, 5 levels, 31x31), one pair: tiled kernels ==
    per-pixel kernels (reference summation order)."""
    import synth
    prev, nxt = synth.frame_pairs_torch(1, 4320, 7680, "cuda", seed=9)
    u, v = gpu_ctx.compute_batch(prev, nxt, 5, 31)
    gpu_ctx.set_option(lkflow.LKFLOW_OPT_IMPL, 1)
    try:
        un, vn = gpu_ctx.compute_batch(prev, nxt, 5, 31)
    finally:
        gpu_ctx.set_option(lkflow.LKFLOW_OPT_IMPL, 0)
    torch.cuda.synchronize()
    assert (u - un).abs().max().item() <= TOL and (v - vn).abs().max().item() <= TOL


@pytest.mark.parametrize("shape", [(448, 640), (1, 1), (2, 3), (5, 200), (37, 9), (1080, 1080)])
def test_edge_map_bit_exact(gpu_ctx, oracle, torch, table_u8, shape):
    """ParallelVideoPyr.cpp:803-813 (threshold TOZERO, Sobel x+y, dilate x2, erode x2): exact integers."""
    rng = np.random.default_rng(shape[0] * 31 + shape[1])
    if shape == (448, 640):
        d = np.stack([table_u8[0], table_u8[1]])
    else:
        d = rng.integers(0, 256, (2,) + shape).astype(np.uint8)
        d[0, : shape[0] // 2] = 3                      # a region below the threshold -> exactly flat
    got = host(gpu_ctx.edge_map(dev(torch, d), 10))
    for k in range(2):
        assert np.array_equal(got[k], oracle.edge_map_u8(d[k], 10))


# ---------------------------------------------------------------- f2 pre-processing (oracle: parity unpinned)
@pytest.mark.parametrize("shape", [(448, 640), (1, 1), (3, 4), (9, 2), (37, 129), (1080, 1080)])
def test_blur_gray_bit_exact(gpu_ctx, oracle, torch, shape):
    """ParallelVideoPyr.cpp:786-787 GaussianBlur(9x9, 1.5) on 8UC3 + BGR2GRAY: integer arithmetic, bit-exact."""
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    f = rng.integers(0, 256, (2,) + shape + (3,)).astype(np.uint8)
    f[1, : shape[0] // 2] = (255, 0, 128)
    got = host(gpu_ctx.blur_gray(dev(torch, f), 9, 1.5))
    for k in range(2):
        assert np.array_equal(got[k], oracle.blur_gray_bgr8(f[k], 9, 1.5))


@pytest.mark.parametrize("ksize,sigma", [(3, 0.8), (5, 1.0), (7, 2.0), (15, 3.0)])
def test_blur_gray_other_kernels(gpu_ctx, oracle, torch, ksize, sigma):
    rng = np.random.default_rng(ksize)
    f = rng.integers(0, 256, (1, 50, 70, 3)).astype(np.uint8)
    assert np.array_equal(host(gpu_ctx.blur_gray(dev(torch, f), ksize, sigma))[0], oracle.blur_gray_bgr8(f[0], ksize, sigma))


def test_blur_gray_argument_errors(gpu_ctx, lkflow, torch):
    f = torch.zeros((1, 8, 8, 3), dtype=torch.uint8, device="cuda")
    for ksize, sigma in [(4, 1.0), (17, 1.0), (9, 0.0), (9, -1.0)]:
        with pytest.raises(lkflow.LKFlowError):
            gpu_ctx.blur_gray(f, ksize, sigma)


def test_frame_diff_bit_exact(gpu_ctx, oracle, torch):
    """ParallelVideoPyr.cpp:800 cur - 0.3*prev (addWeighted float arithmetic): every (cur, prev) byte pair."""
    a, b = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8))
    frames = np.stack([b, a, b])                                   # diffs: a - .3 b, b - .3 a
    got = host(gpu_ctx.frame_diff(dev(torch, frames), 0.3))
    assert np.array_equal(got[0], oracle.frame_diff_u8(a, b, 0.3))
    assert np.array_equal(got[1], oracle.frame_diff_u8(b, a, 0.3))


def test_preprocess_chain_feeds_the_hot_call(gpu_ctx, oracle, torch):
    """BGR frames -> blur/gray -> frame difference -> edge map (ParallelVideoPyr.cpp:786-813) -> flow: the whole
    device-resident video path against the oracle chained the same way."""
    import synth
    h, w, F = 64, 96, 4
    base = synth.video(F, h, w, seed=5)                             # [F,h,w] float64 in [0,1]
    bgr = np.stack([np.clip(base * s, 0, 1) for s in (255, 200, 150)], axis=-1).astype(np.uint8)
    edges = host(gpu_ctx.preprocess_bgr(dev(torch, bgr)))                       # blur/gray + fused diff/edge map
    g = gpu_ctx.blur_gray(dev(torch, bgr))
    assert np.array_equal(host(gpu_ctx.edge_map(gpu_ctx.frame_diff(g, 0.3), 10)), edges)   # unfused == fused
    gray = [oracle.blur_gray_bgr8(bgr[k]) for k in range(F)]
    want = [oracle.edge_map_u8(oracle.frame_diff_u8(gray[k + 1], gray[k], 0.3), 10) for k in range(F - 1)]
    for k in range(F - 1):
        assert np.array_equal(edges[k], want[k])
    U, V = gpu_ctx.compute_video(dev(torch, np.stack(want)), levels=3, win=5)
    for p in range(F - 2):
        uo, vo = oracle.coarse_to_fine(want[p], want[p + 1], 3, 5)
        assert np.abs(host(U[p]) - uo).max() <= TOL and np.abs(host(V[p]) - vo).max() <= TOL       # absolute (ParallelBasicPyr.cpp:787)
    U3, V3 = gpu_ctx.compute_video(dev(torch, np.stack(want)), levels=3, win=3)                     # the reference's window: bit for bit
    for p in range(F - 2):
        uo, vo = oracle.coarse_to_fine(want[p], want[p + 1], 3, 3)
        assert np.array_equal(host(U3[p]), uo) and np.array_equal(host(V3[p]), vo)


def test_video_strict_chain_matches_oracle(gpu_ctx, oracle, torch):
    """ParallelVideoPyr.cpp:832-843 literally: every pair's inputs are overwritten by their level-0 warp and
    the warped `diff` becomes the next pair's `prevDiff` (a serial chain through the whole stream)."""
    import synth
    frames = synth.video(4, 64, 96, seed=11)
    ref = [f.copy() for f in frames]
    want = [oracle.coarse_to_fine(ref[k], ref[k + 1], 3, 5, mutate=True) for k in range(3)]   # mutates ref[k], ref[k+1]
    d = dev(torch, frames)
    U, V = gpu_ctx.compute_video_strict(d, 3, 5)
    for k in range(3):
        assert np.abs(host(U[k]) - want[k][0]).max() <= TOL and np.abs(host(V[k]) - want[k][1]).max() <= TOL
    for k in range(4):                                             # frames end up warped exactly like the oracle's
        assert np.abs(host(d[k]) - ref[k]).max() <= TOL
    Ui, Vi = gpu_ctx.compute_video(dev(torch, frames), 3, 5)       # independent pairs differ from the chain after pair 0
    assert np.array_equal(host(Ui[0]), host(U[0])) and not np.array_equal(host(Ui[1]), host(U[1]))


def test_planes_larger_than_4GiB(gpu_ctx, lkflow, torch):
    """Maximum sizes: a 24000 x 24000 fp64 plane is 4.6 GB, so row/strip addresses do not fit 32-bit byte
    offsets from the plane base.  The marching kernel (strip-relative buffer descriptors) must agree with the
    per-pixel kernel (plain 64-bit pointers) everywhere, in particular in the rows beyond the 4 GiB mark, and
    the 2-level path must run through pyrDown / warp on such planes."""
    n = 24000
    g = torch.Generator(device="cuda").manual_seed(5)
    a = torch.rand((1, n, n), dtype=torch.float64, device="cuda", generator=g)
    b = torch.roll(a, shifts=(1, -1), dims=(1, 2)) * 0.5 + a * 0.5
    u, v = gpu_ctx.lk_batch(a, b, 5)
    gpu_ctx.set_option(lkflow.LKFLOW_OPT_IMPL, 1)
    try:
        un, vn = gpu_ctx.lk_batch(a, b, 5)
    finally:
        gpu_ctx.set_option(lkflow.LKFLOW_OPT_IMPL, 0)
    for got, want in ((u, un), (v, vn)):                            # 576 M noise pixels: a few are ill-conditioned,
        err = (got - want).abs_().div_(want.abs().clamp_(min=1.0))   # so the bar is relative for |flow| > 1
        assert err.max().item() <= TOL
        del err
    assert u[0, -3:].abs().max().item() > 0 and torch.isfinite(u).all().item()
    del un, vn
    U, V = gpu_ctx.compute_batch(a, b, 2, 3)
    assert torch.isfinite(U).all().item() and torch.isfinite(V).all().item()
    lo, hi = U[0, :64], U[0, -64:]                                   # both ends of the plane were written
    assert lo.abs().max().item() > 0 and hi.abs().max().item() > 0


def test_video_batch_shares_pyramids_bitwise(gpu_ctx, torch):
    """Consecutive pairs of one frame buffer (next == prev + one frame) build each interior frame's pyramid once;
    results must be bit-identical to the same pairs passed as unrelated buffers."""
    import synth
    frames = dev(torch, synth.video(6, 96, 160, seed=21))
    Uv, Vv = gpu_ctx.compute_video(frames, 4, 7)                                   # shared-pyramid path
    Ui, Vi = gpu_ctx.compute_batch(frames[:-1].clone(), frames[1:].clone(), 4, 7)  # independent buffers
    assert torch.equal(Uv, Ui) and torch.equal(Vv, Vi)
    strided = torch.zeros((6, 96, 200), dtype=torch.float64, device="cuda")       # padded rows, same video layout
    strided[:, :, :160] = frames
    Us, Vs = gpu_ctx.compute_video(strided[:, :, :160], 4, 7)
    assert torch.equal(Us, Ui) and torch.equal(Vs, Vi)


def test_random_small_shapes_against_oracle(gpu_ctx, oracle, lkflow, torch):
    """Ragged inputs: 60 seeded random (rows, cols, levels, window) combinations, most of them smaller than the
    window / the block / the prefetch depth, pyramid and single-level paths, batched in pairs of different content."""
    rng = np.random.default_rng(2026)
    wins = [3, 3, 5, 7, 9, 15, 21, 27, 31]
    for it in range(60):
        lv_target = int(rng.integers(1, 5))
        h = int(rng.integers(1, 40)) << (lv_target - 1)
        w = int(rng.integers(1, 40)) << (lv_target - 1)
        levels = min(lv_target, lkflow.getMaxLayer(w, h))
        win = wins[int(rng.integers(0, len(wins)))]
        a = rng.random((2, h, w)); b = rng.random((2, h, w))
        a[1, : h // 2] = 0.25; b[1, : h // 2] = 0.25                      # a flat half: det == 0 -> flow exactly 0
        U, V = gpu_ctx.compute_batch(dev(torch, a), dev(torch, b), levels, win)
        U, V = host(U), host(V)
        for p in range(2):
            uo, vo = oracle.coarse_to_fine(a[p], b[p], levels, win)
            scale = np.maximum(1.0, np.maximum(np.abs(uo), np.abs(vo)))
            assert (np.abs(U[p] - uo) / scale).max() <= TOL and (np.abs(V[p] - vo) / scale).max() <= TOL, (it, h, w, levels, win)


@pytest.mark.parametrize("win", [3, 15, 21])
def test_block_boundary_widths_against_oracle(gpu_ctx, oracle, torch, win):
    """Widths just below / at / above what one LK block (64, 128 or 256 lanes minus the 2R halo) and one pyrDown wave
    (62 columns) cover, heights around the strip heights: every output column next to a block seam is checked."""
    rng = np.random.default_rng(win)
    R = win // 2
    widths = sorted({64 - 2 * R + d for d in (-1, 0, 1)} | {128 - 2 * R + d for d in (-1, 0, 1)} |
                    {256 - 2 * R + d for d in (-1, 0, 1)} | {2 * (128 - 2 * R) + d for d in (0, 1)} | {123, 124, 125, 248, 249})
    for w in widths:
        if w < 2:
            continue
        w2 = w + (w & 1)                                               # 2 levels need even sizes
        for h, levels, ww in ((37, 1, w), (46, 2, w2)):
            a, b = rng.random((1, h, ww)), rng.random((1, h, ww))
            U, V = gpu_ctx.compute_batch(dev(torch, a), dev(torch, b), levels, win)
            uo, vo = oracle.coarse_to_fine(a[0], b[0], levels, win)
            scale = np.maximum(1.0, np.maximum(np.abs(uo), np.abs(vo)))
            assert (np.abs(host(U)[0] - uo) / scale).max() <= TOL and (np.abs(host(V)[0] - vo) / scale).max() <= TOL, (w, h, levels)


def test_empty_and_null_inputs_are_rejected(gpu_ctx, lkflow, torch):
    """Empty inputs (0 rows / 0 columns / 0 pairs) and null pointers return LKFLOW_E_ARG from every compute entry point
    (the reference would dereference an empty Mat); nothing is launched and the context stays usable."""
    L = lkflow.lib()
    a = torch.zeros((1, 8, 8), dtype=torch.float64, device="cuda")
    p, h = a.data_ptr(), gpu_ctx._h
    E = lkflow.LKFLOW_E_ARG
    for rows, cols, pairs in ((0, 8, 1), (8, 0, 1), (8, 8, 0), (-1, 8, 1)):
        assert L.lkflow_compute_batch_f64_dev(h, p, p, 512, 512, pairs, rows, cols, 64, p, p, 512, 64, 1, 3, 0, None) == E
        assert L.lkflow_lk_batch_f64_dev(h, p, p, 512, 512, pairs, rows, cols, 64, p, p, 512, 64, 3, None) == E
        assert L.lkflow_pyrdown_f64_dev(h, p, 512, pairs, rows, cols, 64, p, 512, 64, None) == E
        assert L.lkflow_flow_centroid_f64_dev(h, p, p, 512, pairs, rows, cols, 64, 20.0, p, None) == E
        assert L.lkflow_edge_map_u8_dev(h, p, 64, pairs, rows, cols, 8, 10, p, 512, 64, None) == E
        assert L.lkflow_blur_gray_bgr8_dev(h, p, 192, pairs, rows, cols, 24, 9, 1.5, p, 64, 8, None) == E
    assert L.lkflow_compute_batch_f64_dev(h, None, p, 512, 512, 1, 8, 8, 64, p, p, 512, 64, 1, 3, 0, None) == E
    assert L.lkflow_compute_batch_f64_dev(h, p, p, 512, 512, 1, 8, 8, 64, None, p, 512, 64, 1, 3, 0, None) == E
    assert L.lkflow_compute_batch_f64_dev(None, p, p, 512, 512, 1, 8, 8, 64, p, p, 512, 64, 1, 3, 0, None) == E
    assert L.lkflow_u8_to_f64_dev(h, p, p, 0, None) == lkflow.LKFLOW_OK          # n == 0: nothing to do
    u, v = gpu_ctx.compute_batch(a, a.clone(), 1, 3)                              # still usable; flat input -> exactly 0
    assert not u.any().item() and not v.any().item()
