// Micro-benchmark: issue rate of fp64 VALU instructions on gfx950 (cycles per wave-instruction per SIMD) at 1, 2 and 4
// waves per SIMD, for independent and dependent chains of v_add_f64 / v_mul_f64 / v_fma_f64, plus v_rcp_f64.
// Build: hipcc --offload-arch=gfx950 -O3 -o fp64_rate fp64_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP, int NACC>
__global__ void k(double *out, unsigned long long *cyc, int iters, double seed)
{
    double a[NACC];
    for (int i = 0; i < NACC; ++i) a[i] = seed + i + threadIdx.x * 1e-3;
    const double b = seed * 1.0000001, c = seed * 0.5;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 3) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
                if (OP == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(*(float *)&a[i]) : "v"((float)b));
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < NACC; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP, int NACC>
void run(const char *name, int waves_per_simd)
{
    const int iters = 2000, threads = 64 * 4 * waves_per_simd;   // one block per CU, waves spread over the 4 SIMDs
    double *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * threads * 8); hipMalloc(&cyc, 256 * 64 * 8);
    k<OP, NACC><<<256, threads>>>(out, cyc, 10, 1.5);
    k<OP, NACC><<<256, threads>>>(out, cyc, iters, 1.5);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += v; m /= h.size();
    const double ninst = (double)iters * 8 * NACC;
    printf("%-10s acc=%d waves/SIMD=%d : %.2f cycles per instruction per wave, %.2f cycles per instruction per SIMD\n", name, NACC,
           waves_per_simd, m / ninst, m / ninst / waves_per_simd);
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0, 8>("add_f64", w); run<0, 1>("add_f64", w);
        run<1, 8>("mul_f64", w);
        run<2, 8>("fma_f64", w); run<2, 1>("fma_f64", w);
        run<3, 8>("rcp_f64", w);
        run<4, 8>("add_f32", w);
    }
    return 0;
}
