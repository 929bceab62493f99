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
// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (private 4 MiB L2 each), so
// with the natural order vertically adjacent tiles -- which share halo rows -- land on different L2s
// and the halo is fetched from HBM once per XCD.  tile_id() gives every XCD one contiguous run of
// the row-major tile sequence instead (bijective for any tile count).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned xcd_tile_id(unsigned lin, unsigned n)
{
    const unsigned xcd = lin & 7u, q = n >> 3, r = n & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
}

// ------------------------------------------------------------------------------------------------
// cv::pyrDown.  One thread = a 2x2 quad of OUTPUT pixels = a 7x7 source patch read as 16-byte
// pairs (interior) or reflect-101 clamped scalars (border); horizontal [1 4 6 4 1] per source row,
// then vertical, then * 1/256 -- same expression order as OpenCV 2.4 pyrDown_.
// Block = 64 x 4 threads = a 128 x 8 output tile; 1-D grid over (planes, tile rows, tile cols).
// ------------------------------------------------------------------------------------------------
struct PyrDownArgs {
    CPlanes src;
    Planes dst;
    int H, W, Hd, Wd;
    int gx, gy;        // tiles per plane
    int vec_src;       // src: 16-byte aligned base, even pitch/plane -> double2 loads
    int vec_dst;
};

__global__ __launch_bounds__(256) void pyrdown_kernel(PyrDownArgs a)
{
    const unsigned n = gridDim.x;
    unsigned tile = xcd_tile_id(blockIdx.x, n);
    const int bx = tile % a.gx; tile /= a.gx;
    const int by = tile % a.gy;
    const int plane = tile / a.gy;
    const int wrow = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    const int xo = (bx * 64 + (threadIdx.x & 63)) * 2;      // first output column of the quad
    const int yo = (by * 4 + wrow) * 2;                     // first output row (uniform per wave)
    if (yo >= a.Hd) return;
    const int H = a.H, W = a.W;
    const double *s = a.src.at(plane);
    const int pitch = (int)a.src.pitch;
    const int xs = 2 * xo - 2;                              // first source column (even)
    const int ys = 2 * yo - 2;                              // first source row
    // interior tiles (the vast majority) need no border logic at all: decided per wave
    const bool rows_in = ys >= 0 && ys + 6 < H;
    const int xs_lo = 2 * (bx * 64) * 2 - 2, xs_hi = 2 * (bx * 64 + 63) * 2 - 2 + 6;
    const bool wave_fast = a.vec_src && rows_in && xs_lo >= 0 && xs_hi < W;

    double h0[7], h1[7];                                    // horizontal results of the 7 source rows
    if (wave_fast) {
        const double *r = s + ys * pitch + xs;
#pragma unroll
        for (int k = 0; k < 7; ++k, r += pitch) {
            const double2 p0 = *reinterpret_cast<const double2 *>(r);
            const double2 p1 = *reinterpret_cast<const double2 *>(r + 2);
            const double2 p2 = *reinterpret_cast<const double2 *>(r + 4);
            const double v6 = r[6];
            h0[k] = p1.x * 6 + (p0.y + p1.y) * 4 + p0.x + p2.x;
            h1[k] = p2.x * 6 + (p1.y + p2.y) * 4 + p1.x + v6;
        }
    } else {
        if (xo >= a.Wd) return;
        int cx[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) cx[k] = r101(xs + k, W);
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const double *r = s + r101(ys + k, H) * pitch;
            double v[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) v[j] = r[cx[j]];
            h0[k] = v[2] * 6 + (v[1] + v[3]) * 4 + v[0] + v[4];
            h1[k] = v[4] * 6 + (v[3] + v[5]) * 4 + v[2] + v[6];
        }
    }
    const double o00 = (h0[2] * 6 + (h0[1] + h0[3]) * 4 + h0[0] + h0[4]) * (1.0 / 256);
    const double o01 = (h1[2] * 6 + (h1[1] + h1[3]) * 4 + h1[0] + h1[4]) * (1.0 / 256);
    const double o10 = (h0[4] * 6 + (h0[3] + h0[5]) * 4 + h0[2] + h0[6]) * (1.0 / 256);
    const double o11 = (h1[4] * 6 + (h1[3] + h1[5]) * 4 + h1[2] + h1[6]) * (1.0 / 256);
    double *d = a.dst.at(plane) + yo * (int)a.dst.pitch + xo;
    const bool two_c = xo + 1 < a.Wd, two_r = yo + 1 < a.Hd;
    if (xo >= a.Wd) return;
    if (two_c && a.vec_dst) {
        *reinterpret_cast<double2 *>(d) = make_double2(o00, o01);
        if (two_r) *reinterpret_cast<double2 *>(d + a.dst.pitch) = make_double2(o10, o11);
    } else {
        d[0] = o00;
        if (two_c) d[1] = o01;
        if (two_r) {
            d[a.dst.pitch] = o10;
            if (two_c) d[a.dst.pitch + 1] = o11;
        }
    }
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
// ParallelVideoPyr.cpp:646-703.  H, W are the fine level's size (= 2*Hc, 2*Wc, both even).
// One thread = a 2x2 quad of fine pixels, which shares one 3x3 coarse neighbourhood per flow
// plane for cv::pyrUp (branch-free border handling, same roundings as pyrup_at()); 32 independent
// gather loads per thread keep the memory pipe busy; 16-byte stores.
// Block = 64 x 4 threads = a 128 x 8 pixel tile; XCD-aware 1-D grid.
// ------------------------------------------------------------------------------------------------
struct WarpArgs {
    CPlanes img1, img2, cu, cv;
    Planes out1, out2;
    int H, W;
    int gx, gy;
    int vec_out;
};

__device__ __forceinline__ void pyrup_quad(const double *c, int cp, int Hc, int Wc, int i, int j,
                                           double (&o)[2][2])
{
    const int jm = j > 0 ? j - 1 : 0, jp = j < Wc - 1 ? j + 1 : Wc - 1;
    const bool single = Wc == 1, left = j == 0, right = j == Wc - 1;
    const double mB = single ? 8.0 : (right ? 7.0 : 6.0);
    const double wA = (single || left) ? 0.0 : 1.0;
    const double wC = (single || right) ? 0.0 : (left ? 2.0 : 1.0);
    const bool dup = single || right;
    const int s0 = i - 1 < 0 ? min(1, Hc - 1) : i - 1;      // r101(2k, 2Hc)/2
    const int s2 = i + 1 > Hc - 1 ? Hc - 1 : i + 1;
    const int srow[3] = {s0, i, s2};
    double ev[3], od[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double *r = c + srow[k] * cp;
        const double A = r[jm], B = r[j], C = r[jp];
        ev[k] = (A * wA + B * mB) + C * wC;
        od[k] = (B + (dup ? B : C)) * 4;
    }
    o[0][0] = (ev[0] + ev[1] * 6 + ev[2]) * (1.0 / 64);
    o[0][1] = (od[0] + od[1] * 6 + od[2]) * (1.0 / 64);
    o[1][0] = ((ev[1] + ev[2]) * 4) * (1.0 / 64);
    o[1][1] = ((od[1] + od[2]) * 4) * (1.0 / 64);
}

__global__ __launch_bounds__(256) void warp_kernel(WarpArgs a)
{
    const unsigned n = gridDim.x;
    unsigned tile = xcd_tile_id(blockIdx.x, n);
    const int bx = tile % a.gx; tile /= a.gx;
    const int by = tile % a.gy;
    const int p = tile / a.gy;
    const int x = (bx * 64 + (threadIdx.x & 63)) * 2;
    const int y = (by * 4 + (threadIdx.x >> 6)) * 2;
    const int H = a.H, W = a.W;
    if (x >= W || y >= H) return;
    double upu[2][2], upv[2][2];
    pyrup_quad(a.cu.at(p), (int)a.cu.pitch, H >> 1, W >> 1, y >> 1, x >> 1, upu);
    pyrup_quad(a.cv.at(p), (int)a.cv.pitch, H >> 1, W >> 1, y >> 1, x >> 1, upv);
    const double *I1 = a.img1.at(p), *I2 = a.img2.at(p);
    const long long ip = a.img1.pitch;
    double w1[2][2], w2[2][2];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx)
            warp_pair_at(I1, I2, ip, H, W, y + dy, x + dx, upu[dy][dx], upv[dy][dx], w1[dy][dx], w2[dy][dx]);
    double *o1 = a.out1.at(p) + y * (int)a.out1.pitch + x;
    double *o2 = a.out2.at(p) + y * (int)a.out2.pitch + x;
    if (a.vec_out) {
        *reinterpret_cast<double2 *>(o1) = make_double2(w1[0][0], w1[0][1]);
        *reinterpret_cast<double2 *>(o1 + a.out1.pitch) = make_double2(w1[1][0], w1[1][1]);
        *reinterpret_cast<double2 *>(o2) = make_double2(w2[0][0], w2[0][1]);
        *reinterpret_cast<double2 *>(o2 + a.out2.pitch) = make_double2(w2[1][0], w2[1][1]);
    } else {
        o1[0] = w1[0][0]; o1[1] = w1[0][1]; o1[a.out1.pitch] = w1[1][0]; o1[a.out1.pitch + 1] = w1[1][1];
        o2[0] = w2[0][0]; o2[1] = w2[0][1]; o2[a.out2.pitch] = w2[1][0]; o2[a.out2.pitch + 1] = w2[1][1];
    }
}

// ------------------------------------------------------------------------------------------------
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
