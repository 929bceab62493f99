"""GPU suite (-m gpu), round 2 additions: full-size BASELINE configs 2 and 4 against
the oracle, sharded == unsharded on the HIP path, two host threads with two contexts, the 8-bit and asynchronous
host-buffer entry points.  (The frame-kernel placement A/B of round 2 went with its environment knobs: the shipped library
reads no environment variable any more; tools/ab.sh runs such A/Bs against a -DLKFLOW_DEBUG_KNOBS build.)  Everything goes through the C ABI (lkflow_amd.Context)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-6        # ParallelBasicPyr.cpp:787


@pytest.fixture(scope="module")
def torch():
    import torch
    return torch


def _omp_oracle(oracle, a, b, levels, win):
    """oracle/lk_cpu_omp.c mode 1 (every stage row-parallel) on the cores this process may use: bit-identical to
    oracle/lk_oracle.c (tests/test_oracle_golden.py::test_omp_baseline_is_bit_identical_to_oracle), minutes faster."""
    import bench
    oracle.omp_lib().lkc_set_num_threads(bench.usable_cores()[0])
    return oracle.omp_coarse_to_fine(a, b, levels, win, 1)


@pytest.mark.parametrize("rows,cols,levels,win,seed", [(2160, 3840, 5, 21, 7), (4320, 7680, 5, 31, 9), (2448, 3264, 3, 15, 11)],
                         ids=["config2_4k_L5_w21", "config4_8k_L5_w31", "config0_hand_shape_3264x2448_L3_w15"])
def test_full_size_config_vs_oracle(gpu_ctx, oracle, rows, cols, levels, win, seed):
    """BASELINE configs 2 and 4 at FULL size, one pair, against the CPU oracle: pins pyrDown, warp and LK at
    these sizes (the march == per-pixel tests share pyrDown and warp between their two legs).  Config 0's frame
    shape (testdata/hand1.jpg, 3264x2448, L3 w15) at full size too, on synthetic frames: the hand images themselves
    are checked as the committed 816x612 crop (tests/test_gpu_parity.py), a 16 MB fixture would not be "small"."""
    import synth
    a, b = synth.frame_pair(rows, cols, seed=seed)
    u, v = gpu_ctx.computeFlow(a, b, levels, win)
    uo, vo = _omp_oracle(oracle, a, b, levels, win)
    scale = max(1.0, np.abs(uo).max(), np.abs(vo).max())
    du, dv = np.abs(u - uo).max(), np.abs(v - vo).max()
    print(f"{cols}x{rows} L{levels} w{win}: max|du|={du:.3e} max|dv|={dv:.3e} (flow range {scale:.1f} px)")
    assert du <= TOL * scale and dv <= TOL * scale


def test_sharded_stream_equals_unsharded_on_gpu(gpu_ctx, lkflow, torch):
    """SURVEY.md 8(e) "Check" on the HIP path, in one process: the same 9-frame uint8 stream as ONE shard and as
    2 and 3 shards (shard_pairs + stream_flow per shard, what every rank of `bench.py --gpus N` runs) gives
    bit-identical U, V and centroid rows for every pair."""
    import sharding
    import synth
    F, H, W, L, win = 9, 160, 240, 3, 15
    frames = torch.from_numpy(np.clip(np.rint(synth.video(F, H, W, seed=21) * 255.0), 0, 255).astype(np.uint8)).cuda()
    U1, V1, R1 = sharding.stream_flow(frames, gpu_ctx, L, win, thresh=0.5)
    assert U1.shape == (F - 1, H, W) and R1.shape == (F - 1, 3)
    for world in (2, 3, 8):
        seen = 0
        for rank in range(world):
            first, n = lkflow.shard_pairs(F, world, rank)
            if n == 0:
                continue
            U, V, R = sharding.stream_flow(frames[first:first + n].clone(), gpu_ctx, L, win, thresh=0.5)
            assert torch.equal(U, U1[first:first + n - 1]) and torch.equal(V, V1[first:first + n - 1])
            assert torch.equal(R, R1[first:first + n - 1])
            seen += n - 1
        assert seen == F - 1


def test_two_host_threads_two_contexts(lkflow, torch):
    """The header promises "thread-safe per context": two host threads, one context each, concurrently; every
    result equals the single-threaded one bit for bit (the occupancy caches live in the context now)."""
    import synth
    shapes = [(272, 480, 3, 15), (288, 512, 4, 9)]
    inputs = [synth.frame_pairs_torch(3, h, w, "cuda", seed=40 + k) for k, (h, w, _, _) in enumerate(shapes)]
    ref = []
    with lkflow.Context(0) as c:
        for (h, w, lv, win), (p, n) in zip(shapes, inputs):
            ref.append(c.compute_batch(p, n, lv, win))
        torch.cuda.synchronize()
    out, err = [None, None], []

    def work(k):
        try:
            with lkflow.Context(0) as c:
                s = torch.cuda.Stream()
                for _ in range(6):                     # fresh context each thread: first-use occupancy queries race here
                    out[k] = c.compute_batch(inputs[k][0], inputs[k][1], shapes[k][2], shapes[k][3], stream=s)
                s.synchronize()
        except Exception as e:                         # noqa: BLE001
            err.append(e)
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not err, err
    for k in range(2):
        assert torch.equal(out[k][0], ref[k][0]) and torch.equal(out[k][1], ref[k][1])


def test_u8_and_async_host_entry_points(gpu_ctx, lkflow, oracle, table_u8, goldens):
    """lkflow_compute_u8 == lkflow_compute_f64 on convertTo(1/255) of the same frames, bit for bit (and so the
    reference's UPyrTest/VPyrTest at 1e-6); lkflow_enqueue_pair_* + lkflow_wait give the same arrays, with and
    without LKFLOW_OPT_PIN_HOST."""
    a8, b8 = table_u8
    a, b = oracle.u8_to_f64(a8), oracle.u8_to_f64(b8)
    u, v = gpu_ctx.computeFlow(a, b, 7, 3)
    u8, v8 = gpu_ctx.computeFlow_u8(a8, b8, 7, 3)
    assert np.array_equal(u, u8) and np.array_equal(v, v8)
    assert np.abs(u8 - goldens["UPyrTest"]).max() <= TOL and np.abs(v8 - goldens["VPyrTest"]).max() <= TOL
    for pin in (False, True):
        with lkflow.Context(0) as c:
            c.pin_host(pin)
            for _ in range(2):                         # second round re-uses the registered buffers
                ua, va = c.enqueue_pair(a, b, 7, 3)
                c.wait()
                assert np.array_equal(ua, u) and np.array_equal(va, v)
                ub, vb = c.enqueue_pair(a8, b8, 7, 3)
                uc, vc = c.enqueue_pair(a, b, 7, 3)    # a second enqueue first waits for the first
                c.wait()
                assert np.array_equal(ub, u) and np.array_equal(vb, v) and np.array_equal(uc, u) and np.array_equal(vc, v)
            c.pin_host(False)
    # strided 8-bit input (a ROI of a wider frame) and argument errors
    wide = np.zeros((448, 700), dtype=np.uint8)
    wide[:, :640] = a8
    wide2 = np.zeros((448, 700), dtype=np.uint8)
    wide2[:, :640] = b8
    us, vs = gpu_ctx.computeFlow_u8(wide[:, :640], wide2[:, :640], 7, 3)
    assert np.array_equal(us, u) and np.array_equal(vs, v)
    with pytest.raises(lkflow.LKFlowError):
        gpu_ctx.computeFlow_u8(a8, b8, 8, 3)           # more levels than getMaxLayer


def test_two_contexts_pinning_the_same_host_buffers(lkflow, oracle, table_u8, torch):
    """LKFLOW_OPT_PIN_HOST from two contexts on the SAME arrays (a loop that alternates between two engines): results
    unchanged, and a registration one context could not take or release must not stay behind as the thread's sticky
    HIP error (the next torch call would raise it)."""
    a8, b8 = table_u8
    a, b = oracle.u8_to_f64(a8), oracle.u8_to_f64(b8)
    with lkflow.Context(0) as c0:
        ref = c0.computeFlow(a, b, 7, 3)
    outs = [(np.empty_like(a), np.empty_like(a)) for _ in range(2)]
    eng = [lkflow.Context(0) for _ in range(2)]
    try:
        for e in eng:
            e.pin_host(True)
        for k in range(6):
            e = eng[k & 1]
            e.wait()
            e.enqueue_pair(a8 if k % 3 else a, b8 if k % 3 else b, 7, 3, out=outs[k & 1])
        for e in eng:
            e.wait()
        for u, v in outs:
            assert np.array_equal(u, ref[0]) and np.array_equal(v, ref[1])
    finally:
        for e in eng:
            e.pin_host(False)
            e.close()
    x = torch.ones(1024, device="cuda") * 2          # any HIP call of the caller after it
    torch.cuda.synchronize()
    assert float(x.sum()) == 2048.0


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,levels,win", [(136, 240, 3, 15), (272, 480, 3, 9), (96, 300, 2, 5), (540, 960, 3, 13), (272, 480, 3, 21),
                                                (544, 960, 5, 21), (136, 240, 2, 17), (64, 96, 2, 19), (1080, 1920, 3, 15), (272, 480, 3, 23), (272, 480, 3, 25),
                                                (544, 960, 4, 27), (272, 480, 3, 29), (544, 960, 5, 31), (1080, 1920, 3, 31)])
def test_fused_warp_option_is_bit_identical(lkflow, torch, rows, cols, levels, win):
    """LKFLOW_OPT_FUSE_WARP (the warp inside the wave-specialised LK kernel: W role of lk_split.h, frame pixels by
    lk_border_pixel<., true> in trailing workgroups) gives the bits of warp_kernel + lk_split_kernel for every window 5..31: same
    pyrUp / map / remap / window-sum expressions in the same order.  Modes 0 (never), 1 (wherever the shape is taken) and 2
    (default: by level size) must agree bit for bit."""
    import synth
    fr = [synth.frame_pair(rows, cols, seed=11 + k, shift=(1.7 + 0.4 * k, -0.9)) for k in range(3)]
    a = torch.from_numpy(np.ascontiguousarray(np.stack([f[0] for f in fr]))).cuda()
    b = torch.from_numpy(np.ascontiguousarray(np.stack([f[1] for f in fr]))).cuda()
    with lkflow.Context(0) as ctx:
        ctx.set_option(lkflow.LKFLOW_OPT_FUSE_WARP, 0)
        U0, V0 = ctx.compute_batch(a, b, levels, win)          # warp_kernel + lk_split_kernel
        ctx.set_option(lkflow.LKFLOW_OPT_FUSE_WARP, 1)
        U1, V1 = ctx.compute_batch(a, b, levels, win)          # fused wherever the kernel takes the shape
        ctx.set_option(lkflow.LKFLOW_OPT_FUSE_WARP, 2)
        U2, V2 = ctx.compute_batch(a, b, levels, win)          # the default: fused where it was measured faster
        torch.cuda.synchronize()
    assert float(U0.abs().max()) > 0.1                     # the pair really moves
    assert torch.equal(U0, U1) and torch.equal(V0, V1)
    assert torch.equal(U0, U2) and torch.equal(V0, V2)


@pytest.mark.gpu
@pytest.mark.parametrize("src_shape,dst_shape", [((480, 640), (1080, 1080)), ((720, 1280), (1080, 1080)), ((1080, 1080), (1080, 1080)),
                                                  ((30, 40), (37, 101)), ((5, 7), (40, 33)), ((1, 1), (9, 9)), ((2, 3), (3, 2)),
                                                  ((1080, 1920), (540, 960))])
def test_resize_cubic_bit_exact(gpu_ctx, oracle, torch, src_shape, dst_shape):
    """f2 / ParallelVideoPyr.cpp:778 (parity unpinned: no fixture, OpenCV absent): the HIP kernel against the oracle's
    restatement of OpenCV 2.4's 8-bit cubic resize, byte for byte, incl. row tails that take the integer path."""
    rng = np.random.default_rng(src_shape[1] * 7 + dst_shape[0])
    src = rng.integers(0, 256, size=(2,) + src_shape + (3,), dtype=np.uint8)
    got = gpu_ctx.resize_cubic(torch.from_numpy(src).cuda(), *dst_shape).cpu().numpy()
    for k in range(2):
        assert np.array_equal(got[k], oracle.resize_cubic_bgr8(src[k], *dst_shape))
    with pytest.raises(Exception):
        gpu_ctx.resize_cubic(torch.from_numpy(src).cuda(), 0, 5)
