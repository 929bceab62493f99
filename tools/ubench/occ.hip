// occupancy query of the LK kernels on the box (hipcc --offload-arch=gfx950 -std=c++20 -ffp-contract=off)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../optical-flow_amd/csrc/lk_batch.h"
template <typename K> void q(const char *name, K k, int threads)
{
    int n = -1; hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, threads, 0);
    hipFuncAttributes fa{}; hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k));
    printf("%-40s blocks/CU %d (err %d)  regs %d  static LDS %zu\n", name, n, (int)e, fa.numRegs, fa.sharedSizeBytes);
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s: CUs %d, LDS/block %zu, LDS/CU %zu, regs/block %d, max threads/CU %d\n", p.name, p.multiProcessorCount,
           p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock, p.maxThreadsPerMultiProcessor);
    q("lk_batch<15,false,false>", lk::lk_batch_kernel<15, false, false>, 512);
    q("lk_batch<15,true,false>", lk::lk_batch_kernel<15, true, false>, 512);
    q("lk_batch<15,true,true>", lk::lk_batch_kernel<15, true, true>, 512);
    q("lk_split<15,2,true,false>", lk::lk_split_kernel<15, 2, true, false>, 512);
    return 0;
}
