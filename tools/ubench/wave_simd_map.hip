// Which SIMD does physical wave p of a workgroup land on?  One 768- / 832- / 512-thread workgroup per CU (or two), every wave
// records HW_REG_HW_ID (gfx9 layout: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]).
//   hipcc --offload-arch=gfx950 -O2 -o wave_simd_map tools/ubench/wave_simd_map.hip && ./wave_simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned *out, int lds_words)
{
    extern __shared__ unsigned dyn[];
    if (lds_words > 0 && threadIdx.x == 0) dyn[0] = 1;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = id;
    // stay resident for a while so that workgroups really co-reside
    for (int k = 0; k < 20000; ++k) __builtin_amdgcn_s_sleep(1);
    __syncthreads();
}
int main()
{
    for (int nt : {512, 768, 832}) {
        const int nb = 8;
        unsigned *d; hipMalloc(&d, nb * 16 * 4); hipMemset(d, 0xFF, nb * 16 * 4);
        hipLaunchKernelGGL(probe, dim3(nb), dim3(nt), 90000, 0, d, 1);      // 90 KB of LDS: one workgroup per CU
        std::vector<unsigned> h(nb * 16);
        hipMemcpy(h.data(), d, nb * 16 * 4, hipMemcpyDeviceToHost);
        std::printf("threads %d: SIMD of waves 0..%d per workgroup\n", nt, nt / 64 - 1);
        for (int b = 0; b < nb; ++b) {
            std::printf("  wg %d (cu %2u se %u):", b, (h[b * 16] >> 8) & 15, (h[b * 16] >> 13) & 7);
            for (int w = 0; w < nt / 64; ++w) std::printf(" %u", (h[b * 16 + w] >> 4) & 3);
            std::printf("\n");
        }
        hipFree(d);
    }
    return 0;
}
