// lk_kernels.h -- HIP kernels of the dense pyramidal LK path (gfx950 / CDNA4, wave64, fp64).
//
// Kernel inventory (one per stage of SURVEY.md 8(a)); every kernel is batched over blockIdx.z:
//   pyrdown_kernel        a2   cv::pyrDown of n planes
//   pyrup_kernel          a9   cv::pyrUp (stage entry point / tests; the pipeline fuses it)
//   warp_kernel           a9+a10+a11  pyrUp(flow) -> float map -> fixed-point bilinear remap
//   lk_march3_kernel<WIN> a3..a8  derivatives, products, win x win sums, 2x2 solve, += pyrUp(flow)  (lk_march3.h)
//   lk_naive_kernel       a3..a8  per-pixel restatement (cross-check only)
//   u8_to_f64_kernel      a15  convertTo(CV_64F, 1/255)
//   centroid_kernel       f1   |flow| threshold + centroid sums
#pragma once
#include "lk_device.h"

#include <utility>

namespace lk {

// ------------------------------------------------------------------------------------------------
// cv::pyrDown.  One thread per output pixel; horizontal [1 4 6 4 1] on five source rows, then
// vertical, then * 1/256 -- same expression order as OpenCV 2.4 pyrDown_.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pyrdown_kernel(CPlanes src, int H, int W, Planes dst, int Hd,
                                                      int Wd)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= Wd || y >= Hd) return;
    const double *s = src.at(blockIdx.z);
    const int xc = r101(2 * x, W), xl1 = r101(2 * x - 1, W), xr1 = r101(2 * x + 1, W);
    const int xl2 = r101(2 * x - 2, W), xr2 = r101(2 * x + 2, W);
    double row[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const double *r = s + (long long)r101(2 * y + k - 2, H) * src.pitch;
        row[k] = r[xc] * 6 + (r[xl1] + r[xr1]) * 4 + r[xl2] + r[xr2];
    }
    dst.at(blockIdx.z)[(long long)y * dst.pitch + x] =
        (row[2] * 6 + (row[1] + row[3]) * 4 + row[0] + row[4]) * (1.0 / 256);
}

// ------------------------------------------------------------------------------------------------
// cv::pyrUp as a stand-alone stage (tests / API); the pipeline evaluates pyrup_at() inside the
// warp and LK kernels instead of materialising the up-sampled flow.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pyrup_kernel(CPlanes src, int Hc, int Wc, Planes dst)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= 2 * Wc || y >= 2 * Hc) return;
    dst.at(blockIdx.z)[(long long)y * dst.pitch + x] =
        pyrup_at(src.at(blockIdx.z), src.pitch, Hc, Wc, y, x);
}

// ------------------------------------------------------------------------------------------------
// Warp of one pyramid level: up-sample the coarse flow, build the float map, remap both images.
// ParallelVideoPyr.cpp:646-703.  H, W are the fine level's size (= 2*Hc, 2*Wc).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void warp_kernel(CPlanes img1, CPlanes img2, int H, int W,
                                                   CPlanes cu, CPlanes cv, Planes out1, Planes out2)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const int p = blockIdx.z;
    const double upu = pyrup_at(cu.at(p), cu.pitch, H >> 1, W >> 1, y, x);
    const double upv = pyrup_at(cv.at(p), cv.pitch, H >> 1, W >> 1, y, x);
    double w1, w2;
    warp_pair_at(img1.at(p), img2.at(p), img1.pitch, H, W, y, x, upu, upv, w1, w2);
    out1.at(p)[(long long)y * out1.pitch + x] = w1;
    out2.at(p)[(long long)y * out2.pitch + x] = w2;
}

// Arguments of the single-level LK kernels (lk_march3.h, lk_naive_kernel).
struct LKArgs {
    CPlanes i1, i2;   // level images (H x W)
    Planes u, v;      // outputs (H x W)
    CPlanes cu, cv;   // coarse total flow (H/2 x W/2), only if HAS_UP
    int H, W;
    int th;           // output rows per block
};

// ------------------------------------------------------------------------------------------------
// Per-pixel LK (cross-check of the marching kernel on the GPU; reference summation order).
// ------------------------------------------------------------------------------------------------
template <bool HAS_UP>
__global__ __launch_bounds__(256) void lk_naive_kernel(LKArgs a, int win)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int H = a.H, W = a.W;
    if (x >= W || y >= H) return;
    const int pair = blockIdx.z;
    const double *I1 = a.i1.at(pair), *I2 = a.i2.at(pair);
    const int r = win / 2;
    double u = 0.0, v = 0.0;
    if (y >= 1 && y <= H - 2 && x >= 1 && x <= W - 2) {
        double S[5] = {0, 0, 0, 0, 0};
        const bool clipped = (y - r < 0) || (y + r > H - 1) || (x - r < 0) || (x + r > W - 1);
        if (clipped) {
            lk_products_at(I1, I2, a.i1.pitch, H, W, y, x, S);
            for (int f = 0; f < 5; ++f) S[f] = S[f] * (double)(win * win);
        } else {
            for (int i = -r; i <= r; ++i)
                for (int j = -r; j <= r; ++j) {
                    double p[5];
                    lk_products_at(I1, I2, a.i1.pitch, H, W, y + i, x + j, p);
                    for (int f = 0; f < 5; ++f) S[f] += p[f];
                }
        }
        lk_solve(S, u, v);
    }
    if (HAS_UP) {
        u += pyrup_at(a.cu.at(pair), a.cu.pitch, H >> 1, W >> 1, y, x);
        v += pyrup_at(a.cv.at(pair), a.cv.pitch, H >> 1, W >> 1, y, x);
    }
    a.u.at(pair)[(long long)y * a.u.pitch + x] = u;
    a.v.at(pair)[(long long)y * a.v.pitch + x] = v;
}

// ------------------------------------------------------------------------------------------------
// convertTo(CV_64FC1, 1.0/255)  (SerialPyr.cpp:237)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void u8_to_f64_kernel(const unsigned char *src, double *dst,
                                                        size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = (double)src[i] * (1.0 / 255);
}

// ------------------------------------------------------------------------------------------------
// Flow consumer (ParallelVideoPyr.cpp:845-873): count / sum-of-x / sum-of-y of the pixels with
// |flow| >= thresh.  Sums are integers < 2^53, accumulated in fp64 -> exact in any order.
// out[3*pair + {0,1,2}] must be zeroed by the caller (the launcher does it).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void centroid_kernel(CPlanes U, CPlanes V, int H, int W,
                                                       double thresh, double *out)
{
    const int pair = blockIdx.z;
    const double *u = U.at(pair), *v = V.at(pair);
    double c = 0, sx = 0, sy = 0;
    for (int y = blockIdx.y; y < H; y += gridDim.y) {
        for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < W; x += gridDim.x * blockDim.x) {
            const double xf = u[(long long)y * U.pitch + x], yf = v[(long long)y * V.pitch + x];
            const double val = sqrt(xf * xf + yf * yf);
            if (!(val < thresh)) { c += 1.0; sx += (double)x; sy += (double)y; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        c += __shfl_down(c, off, 64);
        sx += __shfl_down(sx, off, 64);
        sy += __shfl_down(sy, off, 64);
    }
    __shared__ double red[3][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = c; red[1][wave] = sx; red[2][wave] = sy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
        double tc = 0, tx = 0, ty = 0;
        for (int w = 0; w < nw; ++w) { tc += red[0][w]; tx += red[1][w]; ty += red[2][w]; }
        if (tc != 0.0) {
            atomicAdd(out + 3 * pair + 0, tc);
            atomicAdd(out + 3 * pair + 1, tx);
            atomicAdd(out + 3 * pair + 2, ty);
        }
    }
}

}  // namespace lk
