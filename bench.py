#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mpixels/s of dense pyramidal LK on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pairs P] [--config 1080p|4k|8k|pacman|webcam]

One "step" = one pass of the hot path (lkflow_compute_batch_f64_dev: pyramids, coarse-to-fine
warp + LK) over one batch of P synthetic frame pairs that are already resident in HBM.
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); frame pairs are sharded, each
rank owns P pairs (weak scaling) and there is NO collective on the data path -- only the
barriers that bracket the timed region.  Started without torch.distributed.run, `--gpus N` spawns
the N ranks itself (fresh child processes, before anything touches a GPU) and relays rank 0's line.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "optical-flow_amd"))

CONFIGS = {
    # name: (cols, rows, levels, win, default pairs/step/GPU)   -- BASELINE.json configs[1..4]
    "1080p": (1920, 1080, 3, 15, 32),
    "4k": (3840, 2160, 5, 21, 8),
    "8k": (7680, 4320, 5, 31, 32),     # 256-frame batch over 8 GPUs = 33 frames / 32 pairs per rank (north_star)
    "pacman": (240, 160, 3, 15, 244),
    # the reference's own shipped loop: 1080x1080 CV_64F, getMaxLayer = 4 levels, 3x3 (ParallelVideoPyr.cpp:782,831)
    "webcam": (1080, 1080, 4, 3, 48),
}
VIDEO_CONFIGS = ("8k", "pacman", "webcam")   # consecutive pairs of one frame sequence (BASELINE configs 3, 4 + the webcam loop)
HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"


def level_sizes(cols, rows, levels):
    out = [(rows, cols)]
    for _ in range(1, levels):
        r, c = out[-1]
        out.append(((r + 1) // 2, (c + 1) // 2))
    return out


def algorithmic_bytes_per_pair(cols, rows, levels, stream_pairs=0):
    """B_alg = sum_{l>=1} 2s(N_{l-1}+N_l) + sum_l s(4N_l + 2N_{l+1}), s = 8.
    stream_pairs = P > 0: the P pairs are consecutive frames of ONE stream, whose P+1 pyramids are built once
    (not 2P): the pyramid term becomes (P+1)/P * s(N_{l-1}+N_l) per pair."""
    n = [r * c for r, c in level_sizes(cols, rows, levels)] + [0]
    s = 8
    frames_per_pair = (stream_pairs + 1) / stream_pairs if stream_pairs else 2.0
    return int(sum(frames_per_pair * s * (n[l - 1] + n[l]) for l in range(1, levels))) + \
        sum(s * (4 * n[l] + 2 * n[l + 1]) for l in range(levels))


def lk0_bytes_per_pair(cols, rows, levels, fused=False):
    """Algorithmic bytes of the dominant kernel (LK at level 0): read 2 images, read the coarse
    u,v (N0/4 each), write u,v  = s(4 N0 + 2 N1).  (The fused 3x3 kernel reads the UN-warped images: same count.)"""
    n = [r * c for r, c in level_sizes(cols, rows, levels)] + [0]
    return 8 * (4 * n[0] + 2 * n[1])


def csrc_digest():
    """Digest of the kernel sources in the tree (__graft_entry__.csrc_digest)."""
    import __graft_entry__
    return __graft_entry__.csrc_digest()


def loaded_lib_digest():
    """Digest of the sources the LOADED liblkflow.so was built from (the tail of lkflow_version()), or None for an A/B
    library (LKFLOW_LIB) or one built without the stamp: committed profile numbers are attached only to the code they
    were measured on."""
    if os.environ.get("LKFLOW_LIB"):
        return None
    import lkflow_amd
    v = lkflow_amd.lib().lkflow_version().decode()
    return v.rsplit("src:", 1)[1] if "src:" in v else None


def _profile_record(workload):
    """The committed rocprofv3 record of one workload ("colsxrowsxpairs"): profiles/pmc_traffic.json holds one object per
    workload under "workloads" (written by tools/pmc_summary.py), each stamped with the digest of the kernel sources it was
    measured on.  {} unless that digest is the tree's AND the loaded library's."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            top = json.load(fh)
    except (OSError, ValueError):
        return {}
    d = top.get("workloads", {}).get(workload, {})
    sha = d.get("_csrc_sha")
    if not sha or sha != csrc_digest() or sha != loaded_lib_digest():
        return {}
    return d


def measured_traffic(kernel_name, cols, rows, pairs):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC pass (profiles/pmc_traffic.json,
    written by tools/prof_pmc.sh for THIS workload and THIS code: the file carries the digest of the kernel sources it was
    measured on), or None -- a stale file after a kernel change must not be reported as a measurement."""
    d = _profile_record(f"{cols}x{rows}x{pairs}")
    v = d.get(kernel_name)
    return int(v["hbm_read_bytes"] + v["hbm_write_bytes"]) if isinstance(v, dict) and "hbm_read_bytes" in v else None


def rocprof_avg_launch_ms(kernel_name, cols, rows, pairs):
    """Mean duration of the dominant kernel's level-0 dispatches in the committed rocprofv3 --kernel-trace of the same
    command (stamped into profiles/pmc_traffic.json by tools/prof_round.sh), or None if the code has changed since."""
    d = _profile_record(f"{cols}x{rows}x{pairs}")
    v = d.get(kernel_name)
    return v.get("rocprof_avg_launch_ms") if isinstance(v, dict) else None


def usable_cores():
    """Cores this process may really run on: the affinity mask, cut by a cgroup CPU quota if one is set
    (omp_get_max_threads() reports the machine's hardware threads, which a container may not own)."""
    n = len(os.sched_getaffinity(0))
    quota = None
    try:                                                     # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:                                                 # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    eff = n if quota is None else max(1, min(n, int(quota + 0.5)))
    return eff, n, quota


# ---------------------------------------------------------------------------------------------------
# one measured configuration (headline and the extra config objects share this)
# ---------------------------------------------------------------------------------------------------
PREWARM_S = float(os.environ.get("LKFLOW_BENCH_PREWARM_S", "0.3"))


def prewarm(torch, step, seconds=None):
    """Untimed: run `step` for a fixed wall time before the W warm-up steps.  Three 2 ms steps do not bring the part to
    its steady state (clocks / power state after the idle data-generation phase): with --warmup 3 the timed region of
    the headline runs 2.5-4 % slower than with --warmup 60 or 200 (profiles/r02_j_warmup_ab.txt), whatever follows.
    The count is reported in the JSON line (`config.prewarm_steps`)."""
    seconds = PREWARM_S if seconds is None else seconds
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        step()
        n += 1
        if n % 4 == 0:
            torch.cuda.synchronize()                     # (the host must not queue seconds of work ahead of the clock)
    torch.cuda.synchronize()
    return n


PACMAN_FIXTURE = os.path.join(ROOT, "tests", "golden", "pacman_240x160_y.npz")


def make_inputs(torch, synth, name, P, rows, cols, dev, seed, real=False):
    if real:
        # BASELINE config 3 on its real data: the luma frames of the reference's pacman.mp4, decoded by optical-flow_amd/h264_ingest.py
        # (committed fixture: /root/reference does not exist on the GPU box), convertTo(CV_64F, 1 / 255) as SerialPyr.cpp:237
        import numpy as np
        y = np.load(PACMAN_FIXTURE)["Y"]
        assert name == "pacman" and y.shape == (P + 1, rows, cols), y.shape
        frames = torch.from_numpy(y).to(dev).to(torch.float64) * (1.0 / 255)
        return frames[:-1], frames[1:]
    if name in VIDEO_CONFIGS:
        frames = synth.video_torch(P + 1, rows, cols, dev, seed=seed)
        return frames[:-1], frames[1:]
    return synth.frame_pairs_torch(P, rows, cols, dev, seed=seed)


def roofline_objects(lkf, name, cols, rows, levels, win, P, pairs_per_launch, lk0_ms, lk0_n, dt, steps):
    lk0_avg_s = lk0_ms / lk0_n / 1e3
    kname = lkf.lk_kernel_name(rows, cols, win, has_up=levels > 1, fused_ok=True, pairs=pairs_per_launch)
    fused = "w3x2f" in kname or "w3x1f" in kname or kname.endswith(", true, true>")   # the warp of the level runs inside this kernel
    lk0_bytes = lk0_bytes_per_pair(cols, rows, levels, fused) * pairs_per_launch
    achieved = lk0_bytes / lk0_avg_s
    b_alg = algorithmic_bytes_per_pair(cols, rows, levels, P if name in VIDEO_CONFIGS else 0)
    path_achieved = b_alg * P * steps / dt          # per GPU
    roof = {
        "kernel": f"{kname} (LK at pyramid level 0" + (", warp fused" if fused else "") +
                  (", incl. the trailing workgroups that write the clipped-window frame" if "lk_split" in kname else "") + ")",
        "bound": "hbm",
        "achieved": round(achieved / 1e9, 2),
        "peak": HBM_PEAK / 1e9,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK, 4),
        # HBM bytes per launch from the PMC counters (separate rocprofv3 --pmc passes, gfx950 x2
        # FETCH_SIZE correction applied; committed under profiles/), null if not collected for this workload
        "traffic": measured_traffic(kname, cols, rows, pairs_per_launch),
        "algorithmic_bytes_per_launch": lk0_bytes,
        "avg_launch_ms": round(lk0_avg_s * 1e3, 4),           # HIP events on the launch stream, this run
        "rocprof_avg_launch_ms": rocprof_avg_launch_ms(kname, cols, rows, pairs_per_launch),   # committed trace, same code
        "launches_timed": lk0_n,
        "pairs_per_launch": pairs_per_launch,
    }
    # the same fraction from the COMMITTED rocprofv3 kernel trace (profiles/*_kernel_trace_by_grid.csv via pmc_traffic.json): the
    # figure a reader can re-derive from profiles/ alone; null until tools/prof_round.sh has been re-run on this code
    # the level-0 STAGE of SURVEY.md's model (warp + LK, "warp fused as gather") is this one launch when the level is fused; with a
    # separate warp_kernel launch (small levels, LKFLOW_MUTATE_INPUTS) the LK launch is only part of it: null here (--profile-all has both)
    roof["level0_stage_ms"] = round(lk0_avg_s * 1e3, 4) if fused else None
    if fused and "lk_split" in kname:
        roof["note"] = ("this launch is the WHOLE level-0 stage of SURVEY.md 8(d)'s model (36 B/px, 'warp fused as gather'): since round 4 the "
                        "warp runs inside the LK kernel.  Rounds 1-3 priced the same 36 B/px against the LK launch alone (r03: 0.475-0.50); the "
                        "r03 figure comparable to this frac is the stage, 36 B/px / (warp 0.549 + LK 0.628 ms) = 0.254")
    rp = roof["rocprof_avg_launch_ms"]
    roof["frac_rocprof"] = round(lk0_bytes / (rp * 1e-3) / HBM_PEAK, 4) if rp else None
    path = {
        "algorithmic_bytes_per_pair": b_alg,
        "achieved_GBps_per_gpu": round(path_achieved / 1e9, 2),
        "frac_of_8TBps": round(path_achieved / HBM_PEAK, 4),
    }
    return roof, path, achieved, path_achieved


def extra_config(lkf, torch, synth, name, device, steps=10, warmup=2, real=False):
    """Extra objects `config_4k` / `config_webcam`: the same measurement as the headline on another
    BASELINE shape, single stream, device-resident, with its own level-0 roofline from HIP events."""
    cols, rows, levels, win, P = CONFIGS[name]
    dev = torch.device("cuda", device)
    prev, nxt = make_inputs(torch, synth, name, P, rows, cols, dev, 4321, real=real)
    U, V = torch.empty_like(prev), torch.empty_like(prev)
    ctx = lkf.Context(device)
    ctx.reserve(rows, cols, levels, P)
    stream = torch.cuda.current_stream()
    prewarm(torch, lambda: ctx.compute_batch(prev, nxt, levels, win, out=(U, V), stream=stream), PREWARM_S / 2)
    for _ in range(warmup):
        ctx.compute_batch(prev, nxt, levels, win, out=(U, V), stream=stream)
    torch.cuda.synchronize()
    ctx.profile(lkf.LKFLOW_STAGE_LK, 0)
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.compute_batch(prev, nxt, levels, win, out=(U, V), stream=stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = ctx.profile_read()
    ctx.profile(False)
    ctx.close()
    lk0_ms, lk0_n = prof[(lkf.LKFLOW_STAGE_LK, 0)]
    roof, path, _, _ = roofline_objects(lkf, name, cols, rows, levels, win, P, P, lk0_ms, lk0_n, dt, steps)
    del prev, nxt, U, V
    torch.cuda.empty_cache()
    return {"workload": f"{cols}x{rows} L{levels} w{win}, {P} " + ("consecutive pairs of one stream" if name in VIDEO_CONFIGS else "independent pairs") +
                        " per step, device-resident, 1 GPU",
            "data": ("pacman.mp4: the 245 luma frames of the reference's file, decoded by optical-flow_amd/h264_ingest.py (fixture "
                     "tests/golden/pacman_240x160_y.npz), convertTo(CV_64F, 1/255)") if real else "synthetic",
            "value": round(P * rows * cols * steps / dt / 1e6, 2), "unit": "Mpixels/s",
            "frame_pairs_per_s": round(P * steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps,
            "roofline": roof, "path_roofline": path,
            "note": "rank 0 only, extra measurement after the timed region (not `value`)"}


def pipelined_rate(lkf, torch, prev, nxt, U, V, levels, win, device, world, chunks=4, steps=20):
    """Extra, NOT the headline: the same batch split over `chunks` contexts/streams that run
    unsynchronised (how a video pipeline keeps several chunks in flight).  Kernel tails of one chunk are
    back-filled by another chunk's kernels, so whole-job throughput is higher, while each LK launch takes
    longer under the contention -- which is why `value`/`roofline` above are taken single-stream."""
    P, rows, cols = prev.shape
    bounds = [P * k // chunks for k in range(chunks + 1)]
    ctxs = [lkf.Context(device) for _ in range(chunks)]
    streams = [torch.cuda.Stream() for _ in range(chunks)]
    for k in range(chunks):
        ctxs[k].reserve(rows, cols, levels, bounds[k + 1] - bounds[k])

    def step():
        for k in range(chunks):
            sl = slice(bounds[k], bounds[k + 1])
            ctxs[k].compute_batch(prev[sl], nxt[sl], levels, win, out=(U[sl], V[sl]), stream=streams[k])
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for c in ctxs:
        c.close()
    return {"chunks_in_flight": chunks, "steps": steps, "rank0_value": round(P * rows * cols * steps / dt / 1e6, 2),
            "unit": "Mpixels/s", "note": "rank 0 only, extra measurement after the timed region"}


def preprocess_rate(ctx, torch, device, frames=33, side=1080, steps=20):
    """Extra, NOT the headline: the video loop's per-frame chain that produces the hot call's inputs
    (ParallelVideoPyr.cpp:786-813) at the reference's 1080 x 1080 working size: blur 9x9 + BGR2GRAY,
    then cur - 0.3*prev fused with threshold + Sobel x+y + dilate x2 + erode x2.  Algorithmic bytes per frame
    pixel: 3+1 (blur/gray) + 2+8 (difference + edge map) = 14 B."""
    g = torch.Generator(device="cpu").manual_seed(7)
    bgr = torch.randint(0, 256, (frames, side, side, 3), dtype=torch.uint8, generator=g).to(f"cuda:{device}")
    for _ in range(3):
        ctx.preprocess_bgr(bgr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.preprocess_bgr(bgr)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # the step in front of it (ParallelVideoPyr.cpp:778): resize(frame, .., Size(1080, 1080), 0, 0, INTER_CUBIC) of 720p frames
    cam = torch.randint(0, 256, (frames, 720, 1280, 3), dtype=torch.uint8, generator=g).to(f"cuda:{device}")
    for _ in range(3):
        ctx.resize_cubic(cam, side, side)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.resize_cubic(cam, side, side)
    torch.cuda.synchronize()
    dr = (time.perf_counter() - t0) / steps
    return {"frames": frames, "shape": [side, side, 3], "ms_per_batch": round(dt * 1e3, 4),
            "frames_per_s": round(frames / dt, 1), "algorithmic_GBps": round(14.0 * frames * side * side / dt / 1e9, 1),
            "resize_cubic": {"from": [720, 1280, 3], "ms_per_batch": round(dr * 1e3, 4), "frames_per_s": round(frames / dr, 1),
                             "algorithmic_GBps": round(3.0 * frames * (720 * 1280 + side * side) / dr / 1e9, 1)},
            "note": "rank 0 only, extra measurement after the timed region; uint8 BGR in -> CV_64F edge images out"}


def video_pipeline_rate(lkf, torch, device, frames=50, side=1080, levels=4, win=3, steps=10):
    """Extra, NOT the headline: the reference's whole per-frame loop after the resize at its own working shape
    (ParallelVideoPyr.cpp:786-873; 1080 x 1080, getMaxLayer = 4 levels, 3 x 3 window), device-resident:
    uint8 BGR frames -> blur/gray -> difference + edge map -> dense pyramidal LK on consecutive edge images
    (independent pairs, batched) -> |flow| >= 20 centroid sums; only 3 doubles per pair leave the GPU."""
    ctx = lkf.Context(device)
    g = torch.Generator(device="cpu").manual_seed(11)
    bgr = torch.randint(0, 256, (frames, side, side, 3), dtype=torch.uint8, generator=g).to(f"cuda:{device}")
    ctx.reserve(side, side, levels, frames - 2)

    def step():
        edges = ctx.preprocess_bgr(bgr)                      # [F-1, H, W] float64
        U, V = ctx.compute_video(edges, levels, win)         # F-2 pairs
        return ctx.flow_centroid(U, V, 20.0)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    out.cpu()
    dt = (time.perf_counter() - t0) / steps
    ctx.close()
    return {"frames": frames, "shape": [side, side, 3], "pyr_levels": levels, "win": win,
            "ms_per_batch": round(dt * 1e3, 3), "frames_per_s": round(frames / dt, 1),
            "note": "rank 0 only, extra measurement after the timed region; includes the 3-scalar D2H copy"}


def copy_ceiling(torch, device, nbytes=2 << 30, reps=10):
    """Extra: the achievable-copy ceiling of this box: a device-to-device copy of 2 GiB, bytes read +
    bytes written per second.  The 8 TB/s `roofline.peak` is the spec figure; this is what a plain copy reaches."""
    a = torch.empty(nbytes // 8, dtype=torch.float64, device=f"cuda:{device}").normal_()
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3)


def host_call_rate(ctx, cols, rows, levels, win, reps=5):
    """Extra: the host-buffer drop-in calls (the cv::Mat entry points): host arrays in, host arrays out, synchronous --
    i.e. the PCIe-inclusive rate of one pair.  Never `value`.  f64 = lkflow_compute_f64 (CV_64F frames, 8 B/px up),
    u8 = lkflow_compute_u8 (8-bit frames, convertTo(1/255) on the device, 1 B/px up); each from pageable memory and
    with LKFLOW_OPT_PIN_HOST (the loop's re-used buffers page-locked in place once: true DMA)."""
    import numpy as np
    import synth
    import mmap
    a_np, b_np = synth.frame_pair(rows, cols, seed=99)
    keep = []

    def fresh(shape, dtype, src=None):
        """An array on its own anonymous mapping -- what a C++ caller's cv::Mat::create / malloc returns for a 16 MB plane.
        (H2D from the arrays NumPy hands back after its own computations, recycled ranges of the Python heap, runs at
        16-23 GB/s on the GPU box instead of the link's 55 GB/s: tools/host_call_py.py, tools/ubench/host_call.cpp and
        profiles/r02_h_host_call.txt -- the same calls from C take 1.4 ms.  That case is reported as numpy_heap_arrays.)"""
        m = mmap.mmap(-1, int(np.prod(shape)) * np.dtype(dtype).itemsize)
        keep.append(m)
        arr = np.frombuffer(m, dtype=dtype).reshape(shape)
        arr[...] = 0 if src is None else src
        return arr
    a, b = fresh((rows, cols), np.float64, a_np), fresh((rows, cols), np.float64, b_np)
    a8 = fresh((rows, cols), np.uint8, np.clip(np.rint(a_np * 255.0), 0, 255).astype(np.uint8))
    b8 = fresh((rows, cols), np.uint8, np.clip(np.rint(b_np * 255.0), 0, 255).astype(np.uint8))
    u, v = fresh((rows, cols), np.float64), fresh((rows, cols), np.float64)

    def timed(x, y):
        ctx.enqueue_pair(x, y, levels, win, out=(u, v)); ctx.wait()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.enqueue_pair(x, y, levels, win, out=(u, v)); ctx.wait()
        dt = (time.perf_counter() - t0) / reps
        return {"ms_per_pair": round(dt * 1e3, 3), "Mpixels_per_s": round(rows * cols / dt / 1e6, 1)}
    out = dict(timed(a, b), bytes_over_pcie_per_pair=4 * rows * cols * 8,
               note="rank 0 only, extra: 2 planes H2D + 2 planes D2H from pageable memory around the same kernels "
                    "(lkflow_enqueue_pair_f64 + lkflow_wait = lkflow_compute_f64)")
    out["u8_in"] = dict(timed(a8, b8), bytes_over_pcie_per_pair=2 * rows * cols + 2 * rows * cols * 8)
    out["numpy_heap_arrays"] = dict(timed(a_np, b_np), note="f64 call again with the input arrays exactly as NumPy's own "
                                    "computation left them on the Python heap (see the docstring of fresh())")
    ctx.pin_host(True)
    try:
        out["pinned"] = {"f64_in": timed(a, b), "u8_in": timed(a8, b8),
                         "note": "LKFLOW_OPT_PIN_HOST: the same (re-used) host arrays page-locked in place"}
    finally:
        ctx.pin_host(False)
    # two engines, pairs alternating between them: pair k's download overlaps pair k+1's upload and kernels
    import lkflow_amd as lkf
    eng = [lkf.Context(ctx.device) for _ in range(2)]
    outs = [(np.empty((rows, cols)), np.empty((rows, cols))) for _ in range(2)]
    try:
        for e in eng:
            e.pin_host(True)
        for k in range(2):
            eng[k].enqueue_pair(a8, b8, levels, win, out=outs[k]); eng[k].wait()
        n = 4 * reps
        t0 = time.perf_counter()
        for k in range(n):
            e = eng[k & 1]
            e.wait()
            e.enqueue_pair(a8, b8, levels, win, out=outs[k & 1])
        for e in eng:
            e.wait()
        dt = (time.perf_counter() - t0) / n
        out["two_engines_u8_pinned"] = {"ms_per_pair": round(dt * 1e3, 3), "Mpixels_per_s": round(rows * cols / dt / 1e6, 1),
                                        "note": "lkflow_enqueue_pair_u8 alternating between two contexts (throughput of a host-side "
                                                "video loop, not the latency of one pair)"}
    finally:
        for e in eng:
            e.pin_host(False)
            e.close()
    return out


def cpu_baseline(cols, rows, levels, win, budget_s=20.0):
    """Times oracle/lk_cpu_omp.c (the restated ParallelVideoPyr.cpp CPU path) on this host's USABLE cores."""
    import synth
    from oracle import oracle as O
    O.build()
    cores, affinity, quota = usable_cores()
    omp_max = O.omp_lib().lkc_num_threads()
    O.omp_lib().lkc_set_num_threads(cores)                # mode 1's row loops use omp_get_max_threads()
    a, b = synth.frame_pair(rows, cols, seed=1234)
    O.omp_coarse_to_fine(a[:64, :64].copy(), b[:64, :64].copy(), 1, 3, 0)   # thread pool warm-up
    res = {}
    for mode, name in ((0, "reference task structure"), (1, "all-cores row-parallel")):
        n, t0 = 0, time.perf_counter()
        while True:
            O.omp_coarse_to_fine(a, b, levels, win, mode)
            n += 1
            dt = time.perf_counter() - t0
            if dt > budget_s * (0.6 if mode == 0 else 0.3) or n >= 96:
                break
        res[mode] = (n * rows * cols / dt / 1e6, n, dt, name)
    return {
        "value": round(res[0][0], 4), "unit": "Mpixels/s", "cores": cores, "kind": "port",
        "sample": f"{res[0][1]} pair(s) of {cols}x{rows} L{levels} w{win} in {res[0][2]:.1f}s with "
                  f"oracle/lk_cpu_omp.c mode 0 (ParallelVideoPyr.cpp's 2/3/5-way tasks + 8-thread row loops)",
        "all_cores_row_parallel_value": round(res[1][0], 4),
        "all_cores_sample": f"{res[1][1]} pair(s) in {res[1][2]:.1f}s, mode 1, {cores} threads",
        "core_accounting": {"usable_cores": cores, "sched_affinity": affinity, "cgroup_cpu_quota": quota,
                            "omp_get_max_threads_before_pinning": omp_max},
    }


# ---------------------------------------------------------------------------------------------------
# N > 1: a real stream through the frame queue: scatter uint8 frames from the ingest rank,
# flow of each rank's consecutive pairs, gather of the 3-scalar centroid rows.  Reported apart from `value`.
# ---------------------------------------------------------------------------------------------------
def stream_sharding_extra(lkf, torch, dist, synth, ctx, name, dev, rank, world, frames_per_rank=0, reps=3):
    import sharding
    cols, rows, levels, win, P = CONFIGS[name]
    per = frames_per_rank or min(P, 16)
    F = per * world + 1
    frames = None
    if rank == 0:
        g = torch.Generator(device="cpu").manual_seed(5)
        base = synth.video_torch(F, rows, cols, dev, seed=77)
        frames = (base * 255.0).round().clamp_(0, 255).to(torch.uint8)
        del base
    t_sc = t_cp = t_ga = 0.0
    rows_root = None
    for it in range(reps + 1):
        torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
        local = sharding.scatter_frames(frames, F, rows, cols, src=0, device=dev)
        torch.cuda.synchronize(); dist.barrier(); t1 = time.perf_counter()
        _, _, cent = sharding.stream_flow(local, ctx, levels, win, keep_flow=False)
        torch.cuda.synchronize(); dist.barrier(); t2 = time.perf_counter()
        rows_root = sharding.gather_rows(cent, F, dst=0)
        torch.cuda.synchronize(); dist.barrier(); t3 = time.perf_counter()
        if it:                                              # first pass = warm-up (communicator set-up, workspace)
            t_sc += t1 - t0; t_cp += t2 - t1; t_ga += t3 - t2
    if rank != 0:
        return None
    npx = (F - 1) * rows * cols
    return {"frames": F, "pairs": F - 1, "shape": [rows, cols], "pyr_levels": levels, "win": win, "reps": reps,
            "scatter_ms": round(t_sc / reps * 1e3, 3), "scatter_GBps": round(F * rows * cols / (t_sc / reps) / 1e9, 2),
            "compute_ms": round(t_cp / reps * 1e3, 3), "compute_Mpixels_per_s": round(npx / (t_cp / reps) / 1e6, 2),
            "gather_ms": round(t_ga / reps * 1e3, 3), "rows_gathered": list(rows_root.shape),
            "end_to_end_Mpixels_per_s": round(npx / ((t_sc + t_cp + t_ga) / reps) / 1e6, 2),
            "note": "strong-scaling stream mode, extra (not `value`): rank 0 holds the uint8 stream; point-to-point "
                    "scatter over RCCL (1-frame overlap per shard), convertTo + flow + fused centroid per rank, "
                    "gather of 3 scalars per pair; each phase bracketed by barriers"}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (this parent has not
    touched a GPU and never will), relay rank 0's JSON line, exit with the worst child status."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = rc or p.wait()
    for line in out.decode().splitlines():                  # ONE JSON line on stdout; anything else a library printed
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")   # there (gloo's connection notes) -> stderr
    sys.stdout.flush()
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=0, help="frame pairs per step per GPU")
    ap.add_argument("--config", default="1080p", choices=sorted(CONFIGS))
    ap.add_argument("--win", type=int, default=0, help="override the config's window (experiments only)")
    ap.add_argument("--real-data", action="store_true",
                    help="--config pacman only: the decoded luma frames of the reference's pacman.mp4 (tests/golden fixture) instead of synthetic frames")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed region (no cpu_baseline, no extra objects): use under rocprofv3")
    ap.add_argument("--profile-all", action="store_true", help="HIP-event time of every stage")
    ap.add_argument("--inflight", type=int, default=1,
                    help="split each step's batch over this many contexts/streams that run unsynchronised "
                         "(kernel tails of one chunk are back-filled by another chunk's kernels)")
    ap.add_argument("--graph", action="store_true",
                    help="capture one step into a HIP graph and replay it (launch-bound small batches); "
                         "the per-kernel HIP-event roofline is then taken from a separate eager pass")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)                               # never returns

    import torch
    import lkflow_amd as lkf
    import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 path on a one-GPU box (all ranks on one device over gloo); never set by the driver
    rehearse = os.environ.get("LKFLOW_BENCH_REHEARSE_ONE_GPU") == "1"
    if rehearse:
        local_rank = 0
    args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dist = None
    # LKFLOW_BENCH_FORCE_PG=1 (tests only, never set by the driver): build the process group for ONE rank as well, so that a
    # one-GPU box executes the RCCL leg of the N > 1 path for real (communicator init, barrier, all_reduce, the frame queue)
    if world > 1 or os.environ.get("LKFLOW_BENCH_FORCE_PG") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        tmo = datetime.timedelta(seconds=300)                # a stuck rank aborts the job instead of hanging the node
        if rehearse:
            dist.init_process_group("gloo", timeout=tmo)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=tmo)

    cols, rows, levels, win, p_default = CONFIGS[args.config]
    win = args.win or win
    P = args.pairs or p_default
    dev = torch.device("cuda", local_rank)

    # synthetic translated-noise input generated in HBM (each rank: its own shard).  BASELINE configs 3 and 4 are
    # STREAMS (consecutive pairs of one frame sequence): P+1 frames, pair p = frames p, p+1; the others are pairs.
    prev, nxt = make_inputs(torch, synth, args.config, P, rows, cols, dev, 1234 + rank, real=args.real_data)
    U = torch.empty_like(prev)
    V = torch.empty_like(prev)
    ctx = lkf.Context(local_rank)
    stream = torch.cuda.current_stream()
    K = max(1, min(args.inflight, P))
    if K == 1:
        ctx.reserve(rows, cols, levels, P)

        def eager_step():
            ctx.compute_batch(prev, nxt, levels, win, out=(U, V), stream=stream)
    else:
        # K independent chunks in flight, one context + stream each ("one HIP stream per
        # in-flight pair"); chunk 0 runs on `ctx` so that its LK launches carry the roofline events
        assert not args.graph, "--graph and --inflight are separate experiments"
        bounds = [P * k // K for k in range(K + 1)]
        ctxs = [ctx] + [lkf.Context(local_rank) for _ in range(K - 1)]
        # (experiment knob: LKFLOW_BENCH_INFLIGHT_PRIO="-1,0" gives the chunks' streams HIP priorities -- a high-priority chunk's
        # one-workgroup-per-CU LK launches then leave only their spare registers to the other chunk's pyramid kernels)
        prios = [int(x) for x in os.environ.get("LKFLOW_BENCH_INFLIGHT_PRIO", "").split(",") if x.strip()]
        streams = [torch.cuda.Stream(priority=prios[k % len(prios)]) if prios else torch.cuda.Stream() for k in range(K)]
        for k in range(K):
            ctxs[k].reserve(rows, cols, levels, bounds[k + 1] - bounds[k])
            streams[k].wait_stream(stream)

        def eager_step():
            for k in range(K):
                sl = slice(bounds[k], bounds[k + 1])
                ctxs[k].compute_batch(prev[sl], nxt[sl], levels, win, out=(U[sl], V[sl]), stream=streams[k])

    step = eager_step
    graph = None
    if args.graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ctx.compute_batch(prev, nxt, levels, win, out=(U, V), stream=side)      # warm (workspace, occupancy cache)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                ctx.compute_batch(prev, nxt, levels, win, out=(U, V), stream=side)
        torch.cuda.current_stream().wait_stream(side)
        step = graph.replay

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    n_prewarm = prewarm(torch, step)
    for _ in range(args.warmup):
        step()
    barrier()
    # dominant kernel (LK at level 0) bracketed by HIP events on its own stream, live in the timed region
    if args.graph:                       # events cannot be recorded by the library inside a replayed graph:
        ctx.profile(lkf.LKFLOW_STAGE_LK, 0)        # take the dominant kernel's time from eager launches
        for _ in range(5):
            eager_step()
        barrier()
        prof_eager = ctx.profile_read()
        ctx.profile(False)
    elif args.profile_all:
        ctx.profile()
    else:
        ctx.profile(lkf.LKFLOW_STAGE_LK, 0)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    prof = prof_eager if args.graph else ctx.profile_read()
    ctx.profile(False)

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    if rank == 0:
        px_per_step = P * rows * cols * world
        value = px_per_step * args.steps / dt / 1e6
        lk0_ms, lk0_n = prof[(lkf.LKFLOW_STAGE_LK, 0)]
        pairs_per_launch = P if K == 1 else P // K          # events are on chunk 0's context
        roof, path, achieved, path_achieved = roofline_objects(lkf, args.config, cols, rows, levels, win, P,
                                                               pairs_per_launch, lk0_ms, lk0_n, dt, args.steps)
        lk0_avg_s = lk0_ms / lk0_n / 1e3
        out = {
            "metric": f"Mpixels/s (dense LK, {levels}-level pyr, {win}x{win} win)",
            "value": round(value, 2),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "pacman.mp4 luma frames (tests/golden/pacman_240x160_y.npz)" if args.real_data else "synthetic",
            "config": {
                "workload": (f"{cols}x{rows} synthetic translated-noise frame stream, {levels} levels, {win}x{win} "
                             f"window, {P + 1} frames = {P} consecutive pairs per step per GPU (device-resident)"
                             if args.config in VIDEO_CONFIGS else
                             f"{cols}x{rows} synthetic translated-noise frame pairs, {levels} levels, "
                             f"{win}x{win} window, {P} independent pairs per step per GPU (device-resident)"),
                "cols": cols, "rows": rows, "pyr_levels": levels, "win": win,
                "pairs_per_step_per_gpu": P,
                "parallelism": f"frame-pair sharding x{world} (no data-path collective)",
                "launch": "hip-graph replay" if args.graph else "eager",
                "prewarm_steps": n_prewarm,                  # untimed, before the W warm-up steps (see prewarm())
                "prewarm_s": PREWARM_S,
                "chunks_in_flight": K,
            },
            "frame_pairs_per_s": round(P * world * args.steps / dt, 2),
            "roofline": roof,
            "path_roofline": path,
        }
        if args.profile_all:
            names = {0: "pyrdown", 1: "warp", 2: "lk"}
            out["stage_ms_per_step"] = {f"{names[s]}_L{l}": round(ms / args.steps, 4)
                                        for (s, l), (ms, n) in sorted(prof.items())}

    # N > 1: the frame-queue mode of north_star as an extra object.  It is the only part of this file that moves data
    # between ranks, and it must never cost the headline line: rank 0's line is complete BEFORE it starts, and a deadline
    # timer on every rank prints that line (rank 0) and ends the process if a rank gets stuck in a send / recv / barrier
    # (RCCL's own watchdog would abort the job after the process-group timeout and the line would be lost).
    if dist is not None and not args.no_extras and os.environ.get("LKFLOW_BENCH_NO_STREAM_EXTRA") != "1":
        import threading

        def give_up():
            if rank == 0:
                out["stream_sharding"] = {"error": "deadline: a rank did not finish the stream extra in time"}
                print(json.dumps(out), flush=True)
            # The headline line above is complete and was measured before this extra started: a frame queue that gets stuck must not
            # turn the whole run into a failed one for the launcher (the line says what happened).  LKFLOW_BENCH_STREAM_EXTRA_FAIL_STATUS=3
            # restores the strict form (every rank exits non-zero) for rehearsals of the watchdog.
            os._exit(int(os.environ.get("LKFLOW_BENCH_STREAM_EXTRA_FAIL_STATUS", "0")))
        wd = threading.Timer(float(os.environ.get("LKFLOW_BENCH_STREAM_EXTRA_DEADLINE_S", "150")), give_up)
        wd.daemon = True
        wd.start()
        del prev, nxt, U, V
        torch.cuda.empty_cache()
        stream_extra = None
        try:
            stream_extra = stream_sharding_extra(lkf, torch, dist, synth, ctx, args.config, dev, rank, world)
        except Exception as e:                               # noqa: BLE001 -- an extra must never cost the headline line
            stream_extra = {"error": f"{type(e).__name__}: {e}"} if rank == 0 else None
        wd.cancel()
        prev = nxt = U = V = None
        if rank == 0 and stream_extra is not None:
            out["stream_sharding"] = stream_extra

    if rank == 0:
        solo = world == 1 and dist is None                  # (a forced one-rank process group ran the stream extra instead)
        if solo and K == 1 and not args.graph and P >= 8 and not args.no_extras:
            out["pipelined"] = pipelined_rate(lkf, torch, prev, nxt, U, V, levels, win, local_rank, world)
        if solo and not args.no_extras:
            del prev, nxt, U, V
            torch.cuda.empty_cache()
            ceil_bps = copy_ceiling(torch, local_rank)
            out["copy_ceiling"] = {"GBps": round(ceil_bps / 1e9, 1), "what": "torch device-to-device copy of 2 GiB (hipMemcpy path), bytes read + written; a hand-written float4 "
                                           "copy kernel reaches 6.29 TB/s on this part (MI355X_MICROARCH.md)",
                                   "lk0_algorithmic_frac": round(achieved / ceil_bps, 4),
                                   "lk0_traffic_frac": (round(roof["traffic"] / lk0_avg_s / ceil_bps, 4)
                                                        if roof["traffic"] else None),
                                   "path_algorithmic_frac": round(path_achieved / ceil_bps, 4)}
            out["host_buffer_call"] = host_call_rate(ctx, cols, rows, levels, win)
            if args.config == "1080p":                       # the other BASELINE shapes, in the driver's record
                out["config_4k"] = extra_config(lkf, torch, synth, "4k", local_rank)
                out["config_webcam"] = extra_config(lkf, torch, synth, "webcam", local_rank)
                out["config_8k"] = extra_config(lkf, torch, synth, "8k", local_rank, steps=5)        # BASELINE config 4's shape
                out["config_pacman"] = extra_config(lkf, torch, synth, "pacman", local_rank)         # BASELINE config 3's shape
                if os.path.exists(PACMAN_FIXTURE):                                                   # ... and its real frames
                    out["config_pacman_real"] = extra_config(lkf, torch, synth, "pacman", local_rank, real=True)
            out["preprocess"] = preprocess_rate(ctx, torch, local_rank)
            out["video_pipeline"] = video_pipeline_rate(lkf, torch, local_rank)
        if solo and not args.no_cpu_baseline and not args.no_extras:
            out["cpu_baseline"] = cpu_baseline(cols, rows, levels, win)
        print(json.dumps(out), flush=True)

    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
