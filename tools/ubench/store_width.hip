// store_width.hip -- does the WIDTH of a lane's store matter for streaming bandwidth on MI355X?  Copy of a 2 GiB fp64 buffer with
// 16-byte loads and (a) 16-byte stores, (b) 8-byte stores of consecutive lanes (what warp_kernel / the LK back end / pyrDown
// issue: one column per lane), (c) 16-byte stores from even lanes only after a DPP exchange; each with default and nt policy.
//   hipcc --offload-arch=gfx950 -O3 -o store_width store_width.hip && ./store_width
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
template <int MODE, int AUX>
__global__ __launch_bounds__(256) void copyk(const double *src, double *dst, long long n_per_block)
{
    // each block copies a contiguous chunk of n_per_block doubles (< 2^28), 256 threads x 2 doubles per iteration
    const double *s = src + (long long)blockIdx.x * n_per_block;
    double *d = dst + (long long)blockIdx.x * n_per_block;
    const __amdgpu_buffer_rsrc_t rs = rsrc(s, (unsigned)(n_per_block * 8)), rd = rsrc(d, (unsigned)(n_per_block * 8));
    const int t = threadIdx.x;
    for (int base = 0; base < (int)n_per_block; base += 512 * 4) {
        u4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (base + k * 512 + 2 * t) * 8, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (MODE == 0) __builtin_amdgcn_raw_buffer_store_b128(v[k], rd, (base + k * 512 + 2 * t) * 8, 0, AUX);
            else if (MODE == 1) {
                // two 8-byte stores per lane, each instruction contiguous across the wave: first doubles of all lanes' pairs cannot be
                // contiguous, so emulate "one column per lane": lane t stores element (base + k*512 + t) and (.. + 256 + t)
                u2 a = {v[k].x, v[k].y}, b = {v[k].z, v[k].w};
                __builtin_amdgcn_raw_buffer_store_b64(a, rd, (base + k * 512 + t) * 8, 0, AUX);
                __builtin_amdgcn_raw_buffer_store_b64(b, rd, (base + k * 512 + 256 + t) * 8, 0, AUX);
            }
        }
    }
}
template <int MODE, int AUX> double run(const double *s, double *d, long long n, const char *name)
{
    const int blocks = 4096; const long long per = n / blocks;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((copyk<MODE, AUX>), dim3(blocks), dim3(256), 0, 0, s, d, per);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((copyk<MODE, AUX>), dim3(blocks), dim3(256), 0, 0, s, d, per);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tbs = 2.0 * n * 8 * 20 / (ms * 1e-3) / 1e12;
    printf("%-44s %.3f ms per copy  %.2f TB/s (read + written)\n", name, ms / 20, tbs);
    return tbs;
}
int main()
{
    const long long n = 1ll << 28;                 // 2 GiB per buffer
    double *s, *d; hipMalloc(&s, n * 8); hipMalloc(&d, n * 8); hipMemset(s, 1, n * 8);
    run<0, 0>(s, d, n, "16-B loads, 16-B stores");
    run<0, 2>(s, d, n, "16-B loads, 16-B stores nt");
    run<1, 0>(s, d, n, "16-B loads, 8-B stores (lane = element)");
    run<1, 2>(s, d, n, "16-B loads, 8-B stores nt (lane = element)");
    hipMemcpy(d, s, n * 8, hipMemcpyDeviceToDevice);
    return 0;
}
