// PCIe copy-rate micro-benchmark: what the host-buffer entry points (lkflow_compute_f64 / _u8) can expect from the link.
// One 1080p CV_64F plane (16.6 MB) each way, pageable / registered / hipHostMalloc memory, 1-D vs 2-D async copies.
//   hipcc --offload-arch=gfx950 -O2 -o pcie_rate pcie_rate.hip && ./pcie_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <malloc.h>
#include <unistd.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const int rows = 1080, cols = 1920, reps = 20;
    const size_t row_b = (size_t)cols * 8, n = row_b * rows;
    void *d; CK(hipMalloc(&d, n));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    void *pageable = aligned_alloc(4096, n), *reg = aligned_alloc(4096, n), *pinned;
    memset(pageable, 1, n); memset(reg, 2, n);
    CK(hipHostRegister(reg, n, hipHostRegisterDefault));
    CK(hipHostMalloc(&pinned, n, hipHostMallocDefault)); memset(pinned, 3, n);
    // memory from the brk heap (what glibc hands NumPy for 16 MB arrays once its dynamic mmap threshold has grown)
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    char *heap = (char *)malloc(n + 4096), *heap_reg = (char *)malloc(n + 4096);
    memset(heap, 4, n + 4096); memset(heap_reg, 5, n + 4096);
    printf("brk heap top %p, heap buffers %p %p\n", sbrk(0), (void *)heap, (void *)heap_reg);
    hipError_t re = hipHostRegister(heap_reg + 832, n, hipHostRegisterDefault);
    printf("hipHostRegister(heap memory): %s\n", hipGetErrorString(re));
    struct { const char *name; void *p; } hosts[] = {{"pageable", pageable}, {"registered", reg}, {"hipHostMalloc", pinned},
                                                      {"heap pageable", heap + 832}, {"heap regist.", heap_reg + 832}};
    for (auto &h : hosts)
        for (int dir = 0; dir < 2; ++dir)
            for (int two_d = 0; two_d < 2; ++two_d) {
                double best = 1e9;
                for (int r = 0; r < reps; ++r) {
                    CK(hipStreamSynchronize(s));
                    const double t0 = now();
                    if (two_d) {
                        if (dir == 0) CK(hipMemcpy2DAsync(d, row_b, h.p, row_b, row_b, rows, hipMemcpyHostToDevice, s));
                        else CK(hipMemcpy2DAsync(h.p, row_b, d, row_b, row_b, rows, hipMemcpyDeviceToHost, s));
                    } else {
                        if (dir == 0) CK(hipMemcpyAsync(d, h.p, n, hipMemcpyHostToDevice, s));
                        else CK(hipMemcpyAsync(h.p, d, n, hipMemcpyDeviceToHost, s));
                    }
                    CK(hipStreamSynchronize(s));
                    const double t = now() - t0;
                    if (t < best) best = t;
                }
                printf("%-14s %s %s  %.3f ms  %.1f GB/s\n", h.name, dir ? "D2H" : "H2D", two_d ? "2D" : "1D", best * 1e3, n / best * 1e-9);
            }
    // both directions at once on two streams (registered memory, 1-D)
    {
        hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        void *d2; CK(hipMalloc(&d2, n));
        double best = 1e9;
        for (int r = 0; r < reps; ++r) {
            CK(hipDeviceSynchronize());
            const double t0 = now();
            CK(hipMemcpyAsync(d, reg, n, hipMemcpyHostToDevice, s));
            CK(hipMemcpyAsync(pinned, d2, n, hipMemcpyDeviceToHost, s2));
            CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
            const double t = now() - t0;
            if (t < best) best = t;
        }
        printf("duplex H2D + D2H (two streams, 1D, pinned)  %.3f ms  %.1f GB/s each way\n", best * 1e3, n / best * 1e-9);
    }
    return 0;
}
