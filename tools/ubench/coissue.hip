// Micro-benchmark: can a gfx950 SIMD issue a VALU instruction of one wave and a SALU / LDS instruction of ANOTHER wave
// in the same cycles?  4 waves per SIMD: waves 0,1 of each SIMD run independent v_add_f64; waves 2,3 run (a) nothing,
// (b) v_add_f64 too, (c) s_add_u32 chains, (d) ds_read_b64.  Reports cycles per VALU instruction per SIMD for the
// VALU waves and cycles per instruction for the other waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(1024) void k(double *out, unsigned long long *cyc, int iters)
{
    __shared__ double lds[2048];
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 1024] = 1.0;
    __syncthreads();
    const int wave = threadIdx.x >> 6;            // 16 waves: wave w sits on SIMD (w % 4) (cyclic placement)
    const bool valu_wave = (wave >> 2) < 2;       // first two waves of every SIMD
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = 1.5 + i + threadIdx.x * 1e-3;
    const double b = 1.0000001;
    unsigned s0 = __builtin_amdgcn_readfirstlane(wave), s1 = 3, s2 = 5, s3 = 7;
    const unsigned addr = (threadIdx.x & 63) * 8;
    double acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (valu_wave || MODE == 1) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    } else if (MODE == 2) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                asm volatile("s_add_u32 %0, %0, %4\n s_add_u32 %1, %1, %4\n s_add_u32 %2, %2, %4\n s_add_u32 %3, %3, %4"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(17u));
    } else if (MODE == 3) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                double v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[i]) : "v"(addr), "n"(i * 512));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(v[i]));
                acc += v[0];
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = acc + (double)(s0 + s1 + s2 + s3); for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int MODE>
void run(const char *name)
{
    const int iters = 1000;
    double *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 256 * 1024 * 8); (void)hipMalloc(&cyc, 256 * 16 * 8);
    k<MODE><<<256, 1024>>>(out, cyc, 10);
    k<MODE><<<256, 1024>>>(out, cyc, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 16);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mv = 0, mo = 0; int nv = 0, no = 0;
    for (size_t i = 0; i < h.size(); ++i) { if (((i % 16) >> 2) < 2) { mv += h[i]; ++nv; } else { mo += h[i]; ++no; } }
    mv /= nv; mo /= no;
    printf("%-28s VALU waves: %.2f cycles per v_add_f64 per wave (2 such waves per SIMD -> %.2f per SIMD); other waves: %.2f cycles per 64 instr-slot iteration\n",
           name, mv / (iters * 64.0), mv / (iters * 64.0) / 2, mo / (iters * 64.0));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    run<0>("others idle");
    run<1>("others v_add_f64 too");
    run<2>("others s_add_u32 chains");
    run<3>("others ds_read_b64");
    return 0;
}
