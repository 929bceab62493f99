import os, sys, time, ctypes
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "optical-flow_amd"))
import lkflow_amd as lkf
import synth
rows, cols = 1080, 1920
a, b = synth.frame_pair(rows, cols, seed=99)
u, v = np.empty((rows, cols)), np.empty((rows, cols))
libc = ctypes.CDLL(None, use_errno=True)
libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
def hp():
    for line in open("/proc/self/smaps_rollup"):
        if line.startswith("AnonHuge"): return line.strip()
def collapse(arr):
    lo = (arr.ctypes.data + (2 << 20) - 1) & ~((2 << 20) - 1)
    hi = (arr.ctypes.data + arr.nbytes) & ~((2 << 20) - 1)
    return libc.madvise(lo, hi - lo, 14), libc.madvise(lo, hi - lo, 25), ctypes.get_errno()
def timed(ctx, x, y, u, v, tag, reps=10):
    ctx.enqueue_pair(x, y, 3, 15, out=(u, v)); ctx.wait()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        ctx.enqueue_pair(x, y, 3, 15, out=(u, v)); ctx.wait()
        ts.append(time.perf_counter() - t0)
    print(f"{tag}: mean {np.mean(ts)*1e3:.3f} ms  min {np.min(ts)*1e3:.3f} ms", flush=True)
mode = sys.argv[1]
print(mode, hp())
if mode == "collapse_first":
    print("collapse:", collapse(a), collapse(b), hp())
with lkf.Context(0) as ctx:
    timed(ctx, a, b, u, v, mode + " pageable f64")
