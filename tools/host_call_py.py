"""Why does bench.py's host_buffer_call (Python) take 3.3 ms where tools/ubench/host_call (C) takes 1.4 ms?  Same calls,
timed from Python with NumPy's own arrays and with arrays carved from 2 MiB-aligned, MADV_HUGEPAGE memory."""
import mmap, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "optical-flow_amd"))
import lkflow_amd as lkf
import synth

rows, cols = 1080, 1920
a, b = synth.frame_pair(rows, cols, seed=99)
print("THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
print("input arrays:", a.dtype, a.flags['C_CONTIGUOUS'], a.ctypes.data % 4096)

def huge_array(shape, dtype=np.float64):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    cap = (n + (4 << 20)) & ~((2 << 20) - 1)
    m = mmap.mmap(-1, cap, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    base = np.frombuffer(m, dtype=np.uint8)
    off = (-base.ctypes.data) % (2 << 20)
    m.madvise(mmap.MADV_HUGEPAGE)
    arr = base[off:off + n].view(dtype).reshape(shape)
    arr[...] = 0
    return arr, m

def timed(ctx, x, y, u, v, tag, reps=10):
    ctx.enqueue_pair(x, y, 3, 15, out=(u, v)); ctx.wait()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        ctx.enqueue_pair(x, y, 3, 15, out=(u, v)); ctx.wait()
        ts.append(time.perf_counter() - t0)
    print(f"{tag}: mean {np.mean(ts)*1e3:.3f} ms  min {np.min(ts)*1e3:.3f} ms", flush=True)

u, v = np.empty((rows, cols)), np.empty((rows, cols))
keep = []
ha, hb, hu, hv = [huge_array((rows, cols)) for _ in range(4)]
keep = [ha[1], hb[1], hu[1], hv[1]]
ha[0][...] = a; hb[0][...] = b
with lkf.Context(0) as ctx:
    timed(ctx, a, b, u, v, "numpy arrays, pageable f64")
    timed(ctx, ha[0], hb[0], hu[0], hv[0], "huge-page arrays, pageable f64")
    timed(ctx, ha[0], hb[0], u, v, "huge-page inputs, numpy outputs")
    timed(ctx, a, b, hu[0], hv[0], "numpy inputs, huge-page outputs")
    ctx.pin_host(True)
    timed(ctx, a, b, u, v, "numpy arrays, PIN_HOST f64")
    timed(ctx, ha[0], hb[0], hu[0], hv[0], "huge-page arrays, PIN_HOST f64")
    ctx.pin_host(False)
    assert np.array_equal(u, hu[0]) and np.array_equal(v, hv[0])

# theory: NumPy's a, b live on recycled heap pages (physically scattered 4 KiB pages: every page its own GPU PTE, the DMA
# engine's translation misses bound the copy); collapsing them into 2 MiB pages (MADV_COLLAPSE, Linux >= 6.1) should fix it
import ctypes
libc = ctypes.CDLL(None, use_errno=True)
libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
def collapse(arr):
    lo = (arr.ctypes.data + (2 << 20) - 1) & ~((2 << 20) - 1)
    hi = (arr.ctypes.data + arr.nbytes) & ~((2 << 20) - 1)
    r1 = libc.madvise(lo, hi - lo, 14)
    r2 = libc.madvise(lo, hi - lo, 25)
    return r1, r2, ctypes.get_errno()
a2, b2 = a.copy(), b.copy()          # recycled heap memory again
print("a2 at", hex(a2.ctypes.data), "b2 at", hex(b2.ctypes.data))
with lkf.Context(0) as ctx:
    timed(ctx, a2, b2, hu[0], hv[0], "numpy copies as inputs (before collapse), huge-page outputs")
    print("collapse:", collapse(a2), collapse(b2))
    timed(ctx, a2, b2, hu[0], hv[0], "numpy copies as inputs (after MADV_COLLAPSE), huge-page outputs")

def vma(addr):
    out, hit = [], False
    for line in open("/proc/self/smaps"):
        f = line.split()
        if "-" in f[0] and len(f) >= 5 and all(c in "0123456789abcdef-" for c in f[0]):
            lo, hi = (int(x, 16) for x in f[0].split("-"))
            hit = lo <= addr < hi
            if hit: out.append(line.strip())
        elif hit and f[0] in ("Size:", "Rss:", "AnonHugePages:", "THPeligible:", "VmFlags:", "Shared_Clean:", "Private_Dirty:", "KernelPageSize:"):
            out.append(" ".join(f))
    return " | ".join(out)
for name, arr in (("a", a), ("b", b), ("a2", a2), ("u", u), ("ha", ha[0])):
    print(name, hex(arr.ctypes.data), vma(arr.ctypes.data))

with lkf.Context(0) as ctx:
    timed(ctx, a, b, hu[0], hv[0], "a, b again (before collapse)")
    print("collapse a, b:", collapse(a), collapse(b))
    timed(ctx, a, b, hu[0], hv[0], "a, b after MADV_COLLAPSE")
print("a", vma(a.ctypes.data))
