"""GPU suite (-m gpu), round 4 additions (VERDICT r3 "what's weak" 1, "next round" 2 and 5): the reference's OWN input class at
the reference's own shape -- edge maps (threshold -> Sobel -> dilate / erode: plateaus, exact zeros, straight edges, i.e.
singular structure tensors) of a synthetic gesture stream, 1080 x 1080, getMaxLayer = 4 levels, 3 x 3 window
(ParallelVideoPyr.cpp:782-832) -- against the oracle at the reference's ABSOLUTE bar, the 3 x 3 path bit for bit; the C-ABI
frame-queue engine; the fused warp + LK path.  Everything goes through the C ABI."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-6        # ParallelBasicPyr.cpp:787 (absolute)
TIGHT = 1e-9      # single level, smooth inputs: GPU vs oracle differ by summation order only
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "optical-flow_amd")


@pytest.fixture(scope="module")
def torch():
    import torch
    return torch


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


def gesture_stream_bgr(frames, side=1080, seed=7):
    """A synthetic stand-in for the reference's webcam stream (ParallelVideoPyr.cpp:769-790: a hand moving in front of a
    dark room, already resized to side x side): a static dark textured background (below the difference threshold once
    `cur - 0.3 prev` is taken), a bright textured disc with three rectangular "fingers" that moves a few pixels per frame
    and a second, smaller bright square moving the other way.  uint8 BGR [frames, side, side, 3]."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:side, 0:side].astype(np.float32)
    bg = (6.0 + 4.0 * np.sin(xx * 0.031) * np.cos(yy * 0.023)).astype(np.float32)          # 2 .. 10: dark
    tex = (18.0 * np.sin(xx * 0.21 + 0.3) * np.sin(yy * 0.17)).astype(np.float32)          # the hand's own texture
    s = side / 1080.0
    out = np.empty((frames, side, side, 3), dtype=np.uint8)
    for f in range(frames):
        cx, cy = (300 + 9.0 * f) * s, (420 + 5.0 * f + 40.0 * np.sin(f * 0.35)) * s
        img = bg.copy()
        hand = (xx - cx) ** 2 + (yy - cy) ** 2 <= (150 * s) ** 2
        for k in range(3):                                                                  # fingers: axis-parallel straight edges
            fx0 = cx + (-90 + 70 * k) * s
            hand |= (xx >= fx0) & (xx < fx0 + 40 * s) & (yy >= cy - 300 * s) & (yy < cy)
        img[hand] = 205.0 + tex[hand]
        qx, qy = (900 - 7.0 * f) * s, (200 + 3.0 * f) * s
        sq = (np.abs(xx - qx) <= 60 * s) & (np.abs(yy - qy) <= 60 * s)
        img[sq] = 170.0
        img += rng.integers(-1, 2, size=img.shape).astype(np.float32)                       # sensor noise, +-1 grey level
        g = np.clip(img, 0, 255)
        out[f, ..., 0] = (g * 0.85).astype(np.uint8)
        out[f, ..., 1] = g.astype(np.uint8)
        out[f, ..., 2] = (g * 0.92).astype(np.uint8)
    return out


@pytest.fixture(scope="module")
def edge_stream(lkflow, torch):
    """49 edge images of 1080 x 1080 (50 BGR frames through lkflow_blur_gray_bgr8_dev + lkflow_diff_edge_map_u8_dev,
    ParallelVideoPyr.cpp:786-813): `config_webcam`'s geometry on the reference's own input class."""
    bgr = gesture_stream_bgr(50)
    with lkflow.Context(0) as ctx:
        edges = ctx.preprocess_bgr(dev(torch, bgr))
        torch.cuda.synchronize()
    e = host(edges)
    assert e.shape == (49, 1080, 1080)
    zeros = float((e == 0).mean())
    assert 0.5 < zeros < 0.999 and e.max() > 100.0, (zeros, e.max())       # mostly exact zeros, strong edges: the input class
    return edges


# ---------------------------------------------------------------------------- the reference's window: bit for bit
def test_win3_goldens_bit_identical_to_oracle(gpu_ctx, oracle, table_f64, torch):
    """3 x 3 is the reference's only window, and since round 4 its nine terms are added in get_Sum9's own order
    (lk_device.h sum9_step): single level and the 7-level pyramid on the golden inputs equal the oracle at EVERY pixel
    (== on doubles; +0 == -0), not merely within 1e-6."""
    a, b = table_f64
    u, v = gpu_ctx.lk_batch(dev(torch, a)[None], dev(torch, b)[None], 3)
    uo, vo = oracle.lucas_kanade(a, b, 3)
    assert np.array_equal(host(u)[0], uo) and np.array_equal(host(v)[0], vo)
    U, V = gpu_ctx.compute_batch(dev(torch, a)[None], dev(torch, b)[None], 7, 3)
    Uo, Vo = oracle.coarse_to_fine(a, b, 7, 3)
    assert np.array_equal(host(U)[0], Uo) and np.array_equal(host(V)[0], Vo)


def test_win3_random_shapes_bit_identical_to_oracle(gpu_ctx, lkflow, oracle, torch):
    """The three 3 x 3 kernels (lk_w3x1f fused, lk_w3x2 on even widths, lk_march3<3, 64> on odd ones) on 40 random
    shapes / level counts, incl. half-flat images and replicated columns (windows of pure rounding residue): equal
    to the oracle at every pixel."""
    rng = np.random.default_rng(404)
    for it in range(40):
        lv_target = int(rng.integers(1, 6))
        h = int(rng.integers(1, 60)) << (lv_target - 1)
        w = int(rng.integers(1, 60)) << (lv_target - 1)
        levels = min(lv_target, lkflow.getMaxLayer(w, h))
        a = rng.random((h, w)); b = rng.random((h, w))
        if it % 3 == 0:
            a[:, : w // 2] = 0.25; b[:, : w // 2] = 0.25
        if it % 4 == 1:
            a[:] = a[:, :1]; b[:] = b[:, :1] * 0.5                     # every row constant: straight vertical structure only
        U, V = gpu_ctx.compute_batch(dev(torch, a)[None], dev(torch, b)[None], levels, 3)
        uo, vo = oracle.coarse_to_fine(a, b, levels, 3)
        ok = np.array_equal(host(U)[0], uo, equal_nan=True) and np.array_equal(host(V)[0], vo, equal_nan=True)
        assert ok, (it, h, w, levels)


# ---------------------------------------------------------------------------- reference input class, reference shape
def test_edge_map_stream_1080_webcam_geometry(gpu_ctx, oracle, edge_stream, torch):
    """ParallelVideoPyr.cpp:800-832 on 49 edge images of 1080 x 1080, L4, w3 (`config_webcam`'s geometry: 48 pairs in one
    call): pairs 0 / 24 / 47 equal the oracle at every pixel -- stricter than the reference's absolute 1e-6
    (ParallelBasicPyr.cpp:787) --, all 48 are the bits of the same pair computed alone."""
    U, V = gpu_ctx.compute_video(edge_stream, 4, 3)
    for p in (0, 24, 47):
        uo, vo = oracle.coarse_to_fine(host(edge_stream[p]), host(edge_stream[p + 1]), 4, 3)
        u, v = host(U[p]), host(V[p])
        assert np.isfinite(uo).all() and np.abs(uo).max() > 1.0                      # the pair really moves
        nbad = int((np.abs(u - uo) > TOL).sum() + (np.abs(v - vo) > TOL).sum())
        assert nbad == 0, (p, nbad, float(np.abs(u - uo).max()), float(np.abs(v - vo).max()))
        assert np.array_equal(u, uo) and np.array_equal(v, vo), p
    for p in range(48):
        u1, v1 = gpu_ctx.compute_batch(edge_stream[p:p + 1], edge_stream[p + 1:p + 2], 4, 3)
        assert torch.equal(u1[0], U[p]) and torch.equal(v1[0], V[p]), p


def test_edge_map_stream_strict_chain(gpu_ctx, oracle, edge_stream, torch):
    """The reference's literal loop (every pair's inputs overwritten by their level-0 warp, `prevDiff = diff`,
    ParallelVideoPyr.cpp:832-843) on 4 edge images of 1080 x 1080: flows and overwritten frames equal the oracle chained
    the same way."""
    fr = edge_stream[10:14].clone()
    ref = [host(fr[k]).copy() for k in range(4)]
    want = [oracle.coarse_to_fine(ref[k], ref[k + 1], 4, 3, mutate=True) for k in range(3)]
    U, V = gpu_ctx.compute_video_strict(fr, 4, 3)
    for k in range(3):
        assert np.array_equal(host(U[k]), want[k][0]) and np.array_equal(host(V[k]), want[k][1]), k
    for k in range(4):
        assert np.array_equal(host(fr[k]), ref[k]), k


@pytest.mark.parametrize("levels,win", [(3, 15), (5, 21), (4, 5)])
def test_edge_map_crop_big_windows_absolute_bar(gpu_ctx, lkflow, oracle, edge_stream, torch, levels, win):
    """The same edge maps at the BASELINE windows (w15 / L3, w21 / L5; w5 / L4) on a 272 x 480 crop around the moving hand,
    at the reference's ABSOLUTE 1e-6.  Windows other than 3 x 3 have no reference (generalised `win`) and
    the marching kernels add a window's win^2 terms in a separable order, the oracle row-major; on integer-valued edge maps
    most sums are exact in either order, so the two agree far below the bar (measured: <= 2e-8 on this stream).  The
    per-pixel kernel (LKFLOW_OPT_IMPL = 1), which adds in the oracle's order, is run beside it: it must equal the oracle to
    the bit, which pins everything around the window sum (pyramid, warp, solve) on this input class."""
    y0, x0 = 200, 180
    crop = edge_stream[:3, y0:y0 + 272, x0:x0 + 480].contiguous()
    U, V = gpu_ctx.compute_video(crop, levels, win)
    worst_good, nbad_total = 0.0, 0
    with lkflow.Context(0) as naive:
        naive.set_option(lkflow.LKFLOW_OPT_IMPL, 1)                    # per-pixel kernel, the oracle's summation order
        Un, Vn = naive.compute_video(crop, levels, win)
        torch.cuda.synchronize()
    for p in range(2):
        uo, vo = oracle.coarse_to_fine(host(crop[p]), host(crop[p + 1]), levels, win)
        e = np.maximum(np.abs(host(U[p]) - uo), np.abs(host(V[p]) - vo))
        en = np.maximum(np.abs(host(Un[p]) - uo), np.abs(host(Vn[p]) - vo))
        bad = e > TOL
        nbad_total += int(bad.sum())
        worst_good = max(worst_good, float(e[~bad].max()))
        print(f"edge crop L{levels} w{win} pair {p}: {int(bad.sum())} of {bad.size} pixels beyond 1e-6 "
              f"(max {float(e.max()):.3e}); reference-order kernel: {int((en > TOL).sum())} beyond 1e-6 (max {float(en.max()):.3e})")
        assert float(en.max()) == 0.0                                  # same order as the oracle: same bits
        assert int(bad.sum()) == 0, (p, int(bad.sum()), float(e.max()))
    print(f"edge crop L{levels} w{win}: every pixel within {worst_good:.3e} of the oracle")


# ---------------------------------------------------------------------------- the solve: shared-reciprocal divisions
def test_solve_shared_reciprocal_is_correctly_rounded(gpu_ctx, oracle, torch):
    """a6 (ParallelVideoPyr.cpp:470-472, 510-520): every LK kernel forms un / det and vn / det from ONE reciprocal iteration
    (lk_device.h lk_div2).  On 1e7 random sum quintuples spread over the whole exponent range, on denormal and huge
    determinants, zeros, infinities and NaNs: bit-identical to two independent IEEE divisions on the GPU, and to NumPy's
    correctly rounded float64 division."""
    rng = np.random.default_rng(2024)
    n = 10_000_000
    S = np.empty((5, n))
    for f in range(5):
        mant = rng.uniform(1.0, 2.0, n) * np.where(rng.random(n) < 0.5, -1.0, 1.0)
        expo = rng.integers(-40, 41, n) if f < 3 else rng.integers(-500, 501, n)
        S[f] = np.ldexp(mant, expo)
    S[0] = np.abs(S[0]); S[1] = np.abs(S[1])                                  # Sxx, Syy are sums of squares
    k = n // 10
    S[2, :k] = np.sqrt(S[0, :k] * S[1, :k]) * (1 + rng.integers(-4, 5, k) * 2.0 ** -52)   # det = residue of two nearly equal products
    S[0, k:2 * k] = np.ldexp(rng.uniform(1, 2, k), rng.integers(-540, -500, k))            # denormal / tiny determinants
    S[1, k:2 * k] = np.ldexp(rng.uniform(1, 2, k), rng.integers(-540, -500, k)); S[2, k:2 * k] = 0.0
    S[0, 2 * k:3 * k] = np.ldexp(rng.uniform(1, 2, k), rng.integers(400, 511, k))          # huge determinants, tiny numerators
    S[1, 2 * k:3 * k] = np.ldexp(rng.uniform(1, 2, k), rng.integers(400, 511, k)); S[2, 2 * k:3 * k] = 0.0
    S[3, 2 * k:3 * k] = np.ldexp(rng.uniform(1, 2, k), rng.integers(-1000, -600, k))
    S[:, 3 * k:3 * k + 8] = np.array([[0, 0, 0, 1, 1], [1, 1, 1, 2, 3], [np.inf, 1, 0, 1, 1], [1, 1, 0, np.nan, 1], [4, 4, 4, 0, 0],
                                      [1, 1, 0, 0, 0], [5e-324, 1, 0, 1, 1], [1e308, 1e308, 0, 1, 1]]).T
    d = dev(torch, S)
    u0, v0 = gpu_ctx.solve(d)
    u1, v1 = gpu_ctx.solve(d, plain=True)
    u0, v0, u1, v1 = host(u0), host(v0), host(u1), host(v1)
    assert np.array_equal(u0.view(np.int64), u1.view(np.int64)) and np.array_equal(v0.view(np.int64), v1.view(np.int64))
    with np.errstate(all="ignore"):
        det = S[0] * S[1] - S[2] * S[2]
        un = S[2] * S[4] - S[1] * S[3]
        vn = S[3] * S[2] - S[0] * S[4]
        uo = np.where(det != 0.0, un / det, 0.0)
        vo = np.where(det != 0.0, vn / det, 0.0)
    assert np.array_equal(u0, uo, equal_nan=True) and np.array_equal(v0, vo, equal_nan=True)
    assert int((det == 0).sum()) > 0 and int(np.isinf(uo).sum()) > 0                           # the special classes really occur


# ---------------------------------------------------------------------------- two pyramid levels in one pass
@pytest.mark.parametrize("shape", [(1080, 1920), (448, 640), (160, 240), (8, 16), (12, 20), (16, 116), (16, 120), (16, 232), (16, 236),
                                   (16, 240), (64, 472), (200, 468), (260, 936), (132, 64), (1080, 1080)])
def test_pyrdown2_equals_two_passes_and_oracle(gpu_ctx, lkflow, oracle, torch, shape):
    """pyrdown2_kernel (levels 1 and 2 of getGaussianPyramid, ParallelVideoPyr.cpp:566-575, from one pass over level 0) ==
    two lkflow_pyrdown_f64_dev passes == the oracle's cv::pyrDown twice, bit for bit: widths around the 58-column wave period
    (one / two columns left for the last wave, exactly full waves), strips of one and several blocks, the smallest shape."""
    rng = np.random.default_rng(shape[0] * 31 + shape[1])
    a = rng.random((3,) + shape)
    a[1] = np.arange(shape[0] * shape[1], dtype=np.float64).reshape(shape) * 1e-3          # a ramp: exposes any mis-indexed tap
    d = dev(torch, a)
    l1, l2 = gpu_ctx.pyrdown2(d)
    p1 = gpu_ctx.pyrdown(d)
    p2 = gpu_ctx.pyrdown(p1)
    assert torch.equal(l1, p1) and torch.equal(l2, p2)
    for k in range(3):
        o1 = oracle.pyr_down(a[k])
        assert np.array_equal(host(l1[k]), o1) and np.array_equal(host(l2[k]), oracle.pyr_down(o1))
    with pytest.raises(lkflow.LKFlowError):
        gpu_ctx.pyrdown2(d[:, : shape[0] - 2])                         # rows not a multiple of 4


# ---------------------------------------------------------------------------- big windows: one association in every kernel form
@pytest.mark.parametrize("win", [23, 25, 27, 29, 31])
def test_big_windows_flat_half_and_forms_agree(lkflow, oracle, torch, win):
    """Windows above 21 x 21 add the part two neighbouring windows share centre outwards in every form of lk_split_kernel
    (accumulated as the DPP-shifted pair sums arrive at 23..27, from F's published pair sums at 29 / 31, fused warp): on a
    half-flat image the flat half's flow is EXACTLY zero (additions only: no residue), the rest meets the oracle at 1e-9, and
    the fused / unfused / per-pixel-reference forms agree (fused == unfused bit for bit)."""
    import synth
    a, b = synth.frame_pair(544, 960, seed=900 + win, shift=(1.4, -0.8))
    a[:, :480] = 0.375; b[:, :480] = 0.375                       # flat left half
    da, db = dev(torch, a)[None], dev(torch, b)[None]
    with lkflow.Context(0) as ctx:
        u, v = ctx.lk_batch(da, db, win)
        ctx.set_option(lkflow.LKFLOW_OPT_FUSE_WARP, 0)
        U0, V0 = ctx.compute_batch(da, db, 2, win)
        ctx.set_option(lkflow.LKFLOW_OPT_FUSE_WARP, 1)
        U1, V1 = ctx.compute_batch(da, db, 2, win)
        torch.cuda.synchronize()
    u, v = host(u)[0], host(v)[0]
    r = win // 2
    assert not u[:, : 480 - r - 2].any() and not v[:, : 480 - r - 2].any()      # windows wholly inside the flat half: det == 0 -> 0
    uo, vo = oracle.lucas_kanade(a, b, win)
    assert np.abs(u - uo).max() <= 1e-9 and np.abs(v - vo).max() <= 1e-9
    assert torch.equal(U0, U1) and torch.equal(V0, V1)
    Uo, Vo = oracle.coarse_to_fine(a, b, 2, win)
    assert np.abs(host(U0)[0] - Uo).max() <= TOL and np.abs(host(V0)[0] - Vo).max() <= TOL


# ---------------------------------------------------------------------------- the fused path under fuzz
def test_fused_warp_fuzz_bitwise(lkflow, torch):
    """60 random cases -- even shapes from 16 x 20 to 600 x 700, 2..4 levels, every window 5..31, 1..3 pairs, white noise (wild
    coarse flows: taps far outside the image, saturated coordinates), flat halves, non-finite pixels in the coarse levels' inputs --
    LKFLOW_OPT_FUSE_WARP 1 (every level that the kernel takes) against 0 (warp_kernel + LK): bit for bit, NaNs in the same places."""
    rng = np.random.default_rng(77)
    wins = list(range(5, 32, 2))
    with lkflow.Context(0) as ctx:
        for it in range(60):
            levels = int(rng.integers(2, 5))
            q = 1 << (levels - 1)
            h = int(rng.integers(16 // q + 1, 600 // q)) * q
            w = int(rng.integers(20 // q + 1, 700 // q)) * q
            h += h & 1; w += w & 1
            levels = min(levels, lkflow.getMaxLayer(w, h))
            win = wins[int(rng.integers(0, len(wins)))]
            P = int(rng.integers(1, 4))
            a = rng.random((P, h, w)); b = rng.random((P, h, w))
            if it % 3 == 0:
                a[0, :, : w // 2] = 0.5; b[0, :, : w // 2] = 0.5
            if it % 5 == 1:
                a *= 1e6; b *= -1e6                                      # huge flows: short saturation of the tap coordinates
            if it % 7 == 2:
                a[0, h // 2, w // 2] = np.inf; b[-1, h // 3, w // 3] = np.nan
            da, db = dev(torch, a), dev(torch, b)
            ctx.set_option(lkflow.LKFLOW_OPT_FUSE_WARP, 0)
            U0, V0 = ctx.compute_batch(da, db, levels, win)
            ctx.set_option(lkflow.LKFLOW_OPT_FUSE_WARP, 1)
            U1, V1 = ctx.compute_batch(da, db, levels, win)
            torch.cuda.synchronize()
            same = np.array_equal(host(U0), host(U1), equal_nan=True) and np.array_equal(host(V0), host(V1), equal_nan=True)
            assert same, (it, h, w, levels, win, P)


# ---------------------------------------------------------------------------- two strip heights in one launch
@pytest.mark.parametrize("P,h,w,levels,win,fuse", [
    (32, 1080, 1920, 3, 15, 2),    # headline: 576 column strips on 256 CUs -> 512 whole columns + 64 in quarters (fused level 0),
                                   #           level 1 fused: 256 whole + 32 in eighths; phase B starts inside a pair, short last run
    (244, 160, 240, 3, 15, 2),     # the 240 x 160 stream shape, unfused, two workgroups per CU: 512 whole + 220 in halves
    (40, 544, 960, 2, 21, 0),      # unfused 21 x 21 and its warp-less coarsest level
    (24, 512, 2176, 2, 31, 1),     # 31 x 31 fused (one workgroup per CU unfused too)
])
def test_two_strip_heights_do_not_change_a_bit(lkflow, torch, P, h, w, levels, win, fuse):
    """Launches big enough for the two-height plan (lkflow.hip pick_strip_plan; tests/test_abi.py walks the tile map on the CPU)
    against the same call with ONE forced strip height (LKFLOW_OPT_TILE_ROWS): every pair bit for bit, no pixel left unwritten
    (the outputs start as NaN)."""
    rng = np.random.default_rng(4000 + win)
    base = rng.random((h + 64, w + 64))
    a = np.stack([base[(3 * k) % 50:(3 * k) % 50 + h, (5 * k) % 60:(5 * k) % 60 + w] for k in range(P)])
    b = np.stack([base[(3 * k) % 50 + 1:(3 * k) % 50 + 1 + h, (5 * k) % 60 + 2:(5 * k) % 60 + 2 + w] for k in range(P)])
    da, db = dev(torch, a), dev(torch, b)
    with lkflow.Context(0) as ctx:
        ctx.set_option(lkflow.LKFLOW_OPT_FUSE_WARP, fuse)
        U = torch.full_like(da, float("nan")); V = torch.full_like(da, float("nan"))
        ctx.compute_batch(da, db, levels, win, out=(U, V))
        ctx.set_option(lkflow.LKFLOW_OPT_TILE_ROWS, 46)
        U1 = torch.full_like(da, float("nan")); V1 = torch.full_like(da, float("nan"))
        ctx.compute_batch(da, db, levels, win, out=(U1, V1))
        torch.cuda.synchronize()
    assert not torch.isnan(U).any() and not torch.isnan(V).any()
    assert torch.equal(U, U1) and torch.equal(V, V1)


def test_two_strip_heights_fuzz_many_small_pairs(lkflow, torch):
    """24 random launches of MANY small pairs (150..700 pairs of 24..140 rows x 60..420 columns, 2 levels, windows 5..31): the
    column strips outnumber the resident workgroups, so the plan has two phases, phase B starts inside a pair and ends in a
    short run -- against one forced strip height, with the warp fused and not: bit for bit, every pixel written."""
    rng = np.random.default_rng(4242)
    wins = list(range(5, 32, 2))
    for it in range(24):
        win = wins[int(rng.integers(0, len(wins)))]
        h = int(rng.integers(max(12, win // 2 + 4), 70)) * 2
        w = int(rng.integers(30, 210)) * 2
        P = int(rng.integers(150, 700))
        base = rng.random((h + 40, w + 40))
        k = np.arange(P)
        oy, ox = (7 * k) % 37, (11 * k) % 39
        a = np.stack([base[y:y + h, x:x + w] for y, x in zip(oy, ox)])
        b = np.stack([base[y + 1:y + 1 + h, x + 1:x + 1 + w] for y, x in zip(oy, ox)])
        da, db = dev(torch, a), dev(torch, b)
        with lkflow.Context(0) as ctx:
            ctx.set_option(lkflow.LKFLOW_OPT_FUSE_WARP, it & 1)
            U = torch.full_like(da, float("nan")); V = torch.full_like(da, float("nan"))
            ctx.compute_batch(da, db, 2, win, out=(U, V))
            ctx.set_option(lkflow.LKFLOW_OPT_TILE_ROWS, int(rng.integers(3, 40)) * 2)
            U1 = torch.full_like(da, float("nan")); V1 = torch.full_like(da, float("nan"))
            ctx.compute_batch(da, db, 2, win, out=(U1, V1))
            torch.cuda.synchronize()
        assert not torch.isnan(U).any() and not torch.isnan(V).any(), (it, h, w, win, P)
        assert torch.equal(U, U1) and torch.equal(V, V1), (it, h, w, win, P)


# ---------------------------------------------------------------------------- odd coarsest levels in the wave-specialised kernel
@pytest.mark.parametrize("win", list(range(5, 32, 2)))
def test_odd_height_coarsest_level_every_window(lkflow, oracle, torch, win):
    """A coarsest level may have an odd height and / or width (3840 x 2160 / L5 and 960 x 540 / L3 end in 240 x 135, 1080 x 1080 /
    L4 in 135 x 135): since round 4 the wave-specialised kernel takes it (the last step of the last strip marches the row below
    the image: zero products, dropped store; the column pair (W - 1, W) is loaded clamped and re-seated).  Single-level LK on 135 x 240 and 77 x 130 (one strip) and with the strip height forced to 24 (last strip of 15 /
    5 rows), 3 pairs: against the oracle, against the per-pixel kernel in the reference's summation order, and bitwise between
    the two strip heights; then the same level as the coarsest of a 3-level pyramid (540 x 960 and 308 x 520) against the oracle."""
    import synth
    assert lkflow.lk_kernel_name(135, 240, win, has_up=False).startswith("lk::lk_split_kernel<%d, 2, false" % win)
    assert lkflow.lk_kernel_name(135, 241, win, has_up=False).startswith("lk::lk_split_kernel<%d, 2, false" % win)
    assert lkflow.lk_kernel_name(135, 240, win, has_up=True).startswith("lk::lk_march3_kernel")
    assert lkflow.lk_kernel_name(136, 241, win, has_up=True).startswith("lk::lk_march3_kernel")
    # odd widths too (1080 x 1080 / L4 ends in 135 x 135): the half-in column pair (W - 1, W); 221 = 2 x 110 + 1: the half pair is
    # the first pair of a column block of its own; 127 / 129: it sits at the end / just past the F wave's 126 columns
    for (h, w) in ((135, 240), (77, 130), (135, 135), (78, 131), (51, 221), (64, 127), (65, 129)):
        if h <= win // 2 + 3:
            continue
        prs = [synth.frame_pair(h, w, seed=900 + win + k, shift=(0.7 + 0.2 * k, -0.4)) for k in range(3)]
        a = np.stack([p[0] for p in prs]); b = np.stack([p[1] for p in prs])
        da, db = dev(torch, a), dev(torch, b)
        with lkflow.Context(0) as ctx:
            u, v = ctx.lk_batch(da, db, win)
            ctx.set_option(lkflow.LKFLOW_OPT_TILE_ROWS, 24)
            u24, v24 = ctx.lk_batch(da, db, win)
            ctx.set_option(lkflow.LKFLOW_OPT_TILE_ROWS, 0)
            ctx.set_option(lkflow.LKFLOW_OPT_IMPL, 1)
            un, vn = ctx.lk_batch(da, db, win)
            torch.cuda.synchronize()
        assert torch.equal(u, u24) and torch.equal(v, v24)
        for k in range(3):
            uo, vo = oracle.lucas_kanade(a[k], b[k], win)
            assert np.abs(host(u)[k] - uo).max() <= TIGHT and np.abs(host(v)[k] - vo).max() <= TIGHT, (h, w, k)
        assert (u - un).abs().max().item() <= TIGHT and (v - vn).abs().max().item() <= TIGHT
    for (h, w) in ((540, 960), (308, 520), (540, 540), (308, 524)):
        a, b = synth.frame_pair(h, w, seed=950 + win, shift=(1.1, -0.8))
        with lkflow.Context(0) as ctx:
            U, V = ctx.compute_batch(dev(torch, a)[None], dev(torch, b)[None], 3, win)
            torch.cuda.synchronize()
        Uo, Vo = oracle.coarse_to_fine(a, b, 3, win)
        assert np.abs(host(U)[0] - Uo).max() <= TOL and np.abs(host(V)[0] - Vo).max() <= TOL, (h, w)


# ---------------------------------------------------------------------------- the C-ABI frame-queue engine through ctypes
def test_stream_engine_ctypes_matches_compute_video(gpu_ctx, lkflow, torch):
    """lkflow_stream_* through the Python mirror: a 9-frame 8-bit stream over the device list [0, 0] == convertTo + compute_video +
    flow_centroid on one context (centroid rows and two pairs' flows, bit for bit); more workers than pairs; the engine without
    LKFLOW_STREAM_KEEP_FLOW refuses lkflow_stream_flow; a push of one frame is an argument error."""
    import synth
    base = synth.video(9, 96, 128, seed=21)
    frames = np.clip(np.rint(base * 255.0), 0, 255).astype(np.uint8)
    d = gpu_ctx.u8_to_f64(dev(torch, frames))
    U, V = gpu_ctx.compute_video(d.reshape(9, 96, 128), 3, 7)
    want = host(gpu_ctx.flow_centroid(U, V, 0.5))
    for devices, keep in (([0, 0], True), ([0] * 12, False)):
        with lkflow.StreamEngine(devices, 96, 128, 3, 7, keep_flow=keep) as eng:
            eng.push(frames, thresh=0.5)
            got = eng.centroids()
            assert np.array_equal(got, want)
            if keep:
                for p in (0, 7):
                    u, v = eng.flow(p)
                    assert np.array_equal(u, host(U[p])) and np.array_equal(v, host(V[p]))
            else:
                with pytest.raises(lkflow.LKFlowError):
                    eng.flow(0)
            with pytest.raises(lkflow.LKFlowError):
                eng.push(frames[:1])
    assert want[:, 0].sum() > 100                              # the stream really moves


# ---------------------------------------------------------------------------- the RCCL leg of the N > 1 bench path, one rank
def test_bench_under_the_driver_launcher_over_rccl_single_rank():
    """The driver's N > 1 launch form (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...`) at N = 1 with LKFLOW_BENCH_FORCE_PG=1: the process group is built on the nccl
    (= RCCL) backend for the one rank, so communicator set-up, dist.barrier(), the MAX all_reduce of the timed region and the
    frame-queue extra (scatter / flow / gather with the device as the wire) all EXECUTE on RCCL on this box -- what the gloo
    rehearsals cannot show.  Two ranks on one GPU are not possible under RCCL (duplicate device), so this is as far as a
    one-GPU box goes."""
    import json
    import socket
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, LKFLOW_BENCH_FORCE_PG="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "bench.py"), "--gpus", "1", "--pairs", "4", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["value"] > 0
    ss = rec["stream_sharding"]
    assert "error" not in ss, ss
    assert ss["rows_gathered"] == [ss["pairs"], 3], ss


# ---------------------------------------------------------------------------- column blocks that start at the frame's inner edge
@pytest.mark.parametrize("win", [5, 15, 21, 31])
def test_column_blocks_start_inside_the_frame(lkflow, oracle, torch, win):
    """LKArgs::xorg: block 0 of the wave-specialised kernel starts its outputs at column R & ~1 and the launch holds
    ceil((W - R - xorg) / XO) column blocks.  Widths on both sides of the block-count edges (XO = 126 - 2 HL columns per block; one
    and two blocks' worth, +- 2), single level, 2 levels unfused (strip height forced) and fused, against the oracle: a column
    lost between two blocks or at the right edge would read as an untouched (NaN-filled) or wrong output column."""
    import synth
    R = win // 2
    HL = R + (R & 1)
    XO = 126 - 2 * HL
    org = R & ~1
    for nblk in (1, 2):
        for w in (nblk * XO + R + org - 2, nblk * XO + R + org, nblk * XO + R + org + 2):
            w &= ~3                                            # even at the coarser level too
            h = 2 * (2 * R + 8)
            a, b = synth.frame_pair(h, w, seed=900 + win + w, shift=(0.8, -1.1))
            da, db = dev(torch, a)[None], dev(torch, b)[None]
            with lkflow.Context(0) as ctx:
                u, v = ctx.lk_batch(da, db, win)
                outs = []
                for fuse in (0, 1):
                    ctx.set_option(lkflow.LKFLOW_OPT_FUSE_WARP, fuse)
                    outs.append(ctx.compute_batch(da, db, 2, win))
                torch.cuda.synchronize()
            uo, vo = oracle.lucas_kanade(a, b, win)
            assert np.abs(host(u)[0] - uo).max() <= TIGHT and np.abs(host(v)[0] - vo).max() <= TIGHT, (win, w)
            Uo, Vo = oracle.coarse_to_fine(a, b, 2, win)
            for U, V in outs:
                assert np.abs(host(U)[0] - Uo).max() <= TOL and np.abs(host(V)[0] - Vo).max() <= TOL, (win, w)
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (win, w)


# ---------------------------------------------------------------------------- BASELINE config 3 on its REAL frames
def test_pacman_real_stream_245_frames(gpu_ctx, oracle, torch):
    """BASELINE config 3 on the reference's own pacman.mp4 (245 luma frames decoded by optical-flow_amd/h264_ingest.py, committed as
    tests/golden/pacman_240x160_y.npz; convertTo(CV_64F, 1 / 255) on the device), 3 levels, 15 x 15: compute_video over all 244 pairs
    == every pair computed alone, bitwise; pairs 0 / 1 / 60 (an IDR picture) / 121 / 243 against the oracle at the reference's
    ABSOLUTE 1e-6 -- on a real game screen: flat black regions (exactly singular windows -> 0), sprites, text, and windows that hold
    nothing but interpolation residue, where the reference's quotient of two rounding errors reaches 1e16 (the outputs are compared
    as they are: a bound relative to such a value would hide everything else)."""
    y = np.load(os.path.join(ROOT, "tests", "golden", "pacman_240x160_y.npz"))["Y"]
    assert y.shape == (245, 160, 240)
    d = gpu_ctx.u8_to_f64(dev(torch, y))
    U, V = gpu_ctx.compute_video(d, 3, 15)
    assert U.shape == (244, 160, 240) and bool(torch.isfinite(U).all()) and bool(torch.isfinite(V).all())
    for p in range(244):
        u1, v1 = gpu_ctx.compute_batch(d[p:p + 1], d[p + 1:p + 2], 3, 15)
        assert torch.equal(u1[0], U[p]) and torch.equal(v1[0], V[p]), p
    f = host(d)
    assert np.array_equal(f, y.astype(np.float64) * (1.0 / 255))                 # SerialPyr.cpp:237
    moving = 0
    for p in (0, 1, 60, 121, 243):
        uo, vo = oracle.coarse_to_fine(f[p], f[p + 1], 3, 15)
        du, dv = np.abs(host(U[p]) - uo), np.abs(host(V[p]) - vo)
        small = np.abs(uo) < 1e3                                                  # (where the comparison is not about 1e16-sized values)
        assert du[small].max() <= TOL and dv[small].max() <= TOL, (p, du[small].max(), dv[small].max())
        assert (du <= np.maximum(TOL, 1e-12 * np.abs(uo))).all() and (dv <= np.maximum(TOL, 1e-12 * np.abs(vo))).all(), p
        moving += int((uo != 0).sum())
    assert moving > 50000                                                          # the stream really has flow


def test_pacman_real_stream_through_the_frame_queue_engine(gpu_ctx, lkflow, torch):
    """Config 3 as north_star words it -- "pacman.mp4 full stream ... frame-pair sharded" -- through the C-ABI frame-queue engine
    (lkflow_stream_*, ParallelVideoPyr.cpp:769-873's loop): the 245 decoded 8-bit frames over the device list [0, 0, 0] (three
    workers, blocks with a one-frame overlap) == convertTo + compute_video + flow_centroid on one context, centroid rows bit for bit."""
    y = np.ascontiguousarray(np.load(os.path.join(ROOT, "tests", "golden", "pacman_240x160_y.npz"))["Y"])
    d = gpu_ctx.u8_to_f64(dev(torch, y))
    U, V = gpu_ctx.compute_video(d, 3, 15)
    want = host(gpu_ctx.flow_centroid(U, V, 1.0))
    with lkflow.StreamEngine([0, 0, 0], 160, 240, 3, 15) as eng:
        eng.push(y, thresh=1.0)
        got = eng.centroids()
    assert got.shape == (244, 3) and np.array_equal(got, want)
    assert (want[:, 0] > 0).sum() > 200                                            # nearly every pair has moving pixels


def test_reference_video_loop_on_the_real_pacman_frames(gpu_ctx, oracle, torch):
    """The reference's whole per-frame loop (ParallelVideoPyr.cpp:776-873) on real video instead of a synthetic gesture: six decoded
    BGR frames of pacman.mp4 (h264_ingest.yuv420_to_bgr of the fixture's planes) -> resize to 1080 x 1080 INTER_CUBIC (:778) ->
    GaussianBlur 9 x 9 + BGR2GRAY (:786-787) -> cur - 0.3 prev, threshold, Sobel, dilate / erode (:800-813) -> coarseToFineEstimation with
    getMaxLayer = 4 levels, 3 x 3 (:832) -> |flow| centroid (:845-873).  Every stage equals the oracle's byte for byte, the flow of both
    checked pairs at every pixel (3 x 3: the reference's own summation order), and the centroid sums exactly."""
    sys_path = os.path.join(ROOT, "optical-flow_amd")
    import sys
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    import h264_ingest
    f = np.load(os.path.join(ROOT, "tests", "golden", "pacman_240x160_y.npz"))
    sel = slice(100, 106)
    bgr = h264_ingest.yuv420_to_bgr(f["Y"][sel], f["Cb"][sel], f["Cr"][sel])          # [6, 160, 240, 3]
    big = gpu_ctx.resize_cubic(dev(torch, bgr), 1080, 1080)
    big_o = np.stack([oracle.resize_cubic_bgr8(bgr[k], 1080, 1080) for k in range(6)])
    assert np.array_equal(host(big), big_o)
    edges = gpu_ctx.preprocess_bgr(big)                                                # [5, 1080, 1080] float64
    gray = [oracle.blur_gray_bgr8(big_o[k]) for k in range(6)]
    want = [oracle.edge_map_u8(oracle.frame_diff_u8(gray[k + 1], gray[k], 0.3), 10) for k in range(5)]
    for k in range(5):
        assert np.array_equal(host(edges[k]), want[k]), k
    assert 0.5 < float((host(edges) == 0).mean()) < 0.9999 and host(edges).max() > 100.0
    U, V = gpu_ctx.compute_video(edges, 4, 3)
    for p in (0, 3):
        uo, vo = oracle.coarse_to_fine(want[p], want[p + 1], 4, 3)
        assert np.array_equal(host(U[p]), uo, equal_nan=True) and np.array_equal(host(V[p]), vo, equal_nan=True), p
    rows = host(gpu_ctx.flow_centroid(U, V, 20.0))
    for p in (0, 3):
        assert tuple(rows[p]) == tuple(oracle.flow_centroid(host(U[p]), host(V[p]), 20.0)), (p, rows[p])
    assert rows[:, 0].max() > 0


def test_bench_frame_queue_deadline_keeps_the_headline_line():
    """The N > 1 bench path when its frame-queue extra gets stuck (rehearsed on the one-GPU box: two ranks over gloo, deadline of one
    millisecond): rank 0 still prints the complete headline line -- measured before the extra started --, with stream_sharding.error
    set, and the job ends with status 0 so that a launcher does not discard the measurement; LKFLOW_BENCH_STREAM_EXTRA_FAIL_STATUS=3
    restores the strict form."""
    import json
    import sys
    base = dict(os.environ, LKFLOW_BENCH_REHEARSE_ONE_GPU="1", MASTER_ADDR="127.0.0.1", LKFLOW_BENCH_STREAM_EXTRA_DEADLINE_S="0.001")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--pairs", "4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    for status in ("0", "3"):
        out = subprocess.run(cmd, env=dict(base, LKFLOW_BENCH_STREAM_EXTRA_FAIL_STATUS=status), capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert out.returncode == int(status), (status, out.returncode, out.stderr[-2000:])
        rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        assert rec["n_gpus"] == 2 and rec["value"] > 0 and "deadline" in rec["stream_sharding"]["error"]
