"""Does the 256 MB Infinity Cache help a producer -> consumer pair of bandwidth-bound kernels?  Device-to-device copy rate
(read + written bytes) against the working-set size, and a write-then-read pair (fill, then sum) of the same buffer."""
import time, torch
dev = torch.device("cuda", 0)
for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
    n = mb * (1 << 20) // 8
    x = torch.rand(n, dtype=torch.float64, device=dev); y = torch.empty_like(x)
    for _ in range(3): y.copy_(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = max(5, 4096 // mb)
    for _ in range(reps): y.copy_(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    # producer -> consumer on ONE buffer: y = x * 2 (writes y), then z = y + 1 (reads y back)
    z = torch.empty_like(x)
    for _ in range(3): torch.mul(x, 2.0, out=y); torch.add(y, 1.0, out=z)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): torch.mul(x, 2.0, out=y); torch.add(y, 1.0, out=z)
    torch.cuda.synchronize(); dt2 = (time.perf_counter() - t0) / reps
    print(f"{mb:5d} MB per buffer: copy {2 * n * 8 / dt / 1e9:7.1f} GB/s   mul->add pair {4 * n * 8 / dt2 / 1e9:7.1f} GB/s (2 reads + 2 writes)", flush=True)
    del x, y, z
