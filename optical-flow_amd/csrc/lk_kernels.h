// lk_kernels.h -- HIP kernels of the dense pyramidal LK path (gfx950 / CDNA4, wave64, fp64).
//
// Kernel inventory (one per stage of SURVEY.md 8(a)); every kernel is batched over blockIdx.z:
//   pyrdown_kernel        a2   cv::pyrDown of n planes (row-marching, DPP neighbour taps)
//   pyrup_kernel          a9   cv::pyrUp (stage entry point / tests; the pipeline fuses it)
//   warp_kernel           a9+a10+a11  pyrUp(flow) -> float map -> fixed-point bilinear remap (1x4 strips)
//   lk_split_kernel<WIN>  a3..a8  the same, wave-specialised (front / window-sum / back waves): windows 5..31  (lk_split.h)
//   lk_border_kernel      a5      clipped-window pixels (m(y,x) * win^2) after lk_split_kernel  (lk_split.h)
//   lk_march3_kernel<WIN> a3..a8  derivatives, products, win x win sums, 2x2 solve, += pyrUp(flow)  (lk_march3.h)
//   lk_w3x2_kernel        a3..a8  3 x 3 windows, two columns per lane, no LDS  (lk_w3x2.h)
//   lk_w3x2f_kernel       a9..a12 + a3..a8  3 x 3 windows: warp fused into the LK kernel  (lk_w3x2f.h)
//   lk_naive_kernel       a3..a8  per-pixel restatement (cross-check only)
//   blur_gray_kernel      f2   GaussianBlur(k x k, sigma) on 8UC3 + BGR2GRAY (OpenCV 2.4 fixed point)
//   frame_diff_kernel     f2   cur - rate*prev on 8UC1 (addWeighted float arithmetic)
//   edge_map_kernel       f2   threshold + Sobel x+y + dilate x2 + erode x2 (uint8 diff -> CV_64F input)
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
// cv::pyrDown, row-marching -- same expression order as OpenCV 2.4 pyrDown_:
//   row[x] = s[2x]*6 + (s[2x-1] + s[2x+1])*4 + s[2x-2] + s[2x+2]         (horizontal, per source row)
//   dst[x] = (row2*6 + (row1 + row3)*4 + row0 + row4) * (1/256)            (vertical)
// Lane l of a wave owns the source column PAIR (2*ox, 2*ox+1) of output column ox = ox_w + l - 1 and
// loads it with one 16-byte access: a wave reads 1 KiB contiguous per source row, every line exactly
// once.  The other three horizontal taps come from the neighbouring lanes through DPP wave shifts
// (lanes 0 and 63 are halo lanes, so a wave yields 62 columns); the vertical pass keeps the last
// five horizontal results in registers while the wave marches down its strip of output rows.
// No LDS, no barriers, ~30 VGPRs.  Block = 4 waves side by side (248 output columns).
// ------------------------------------------------------------------------------------------------
#ifndef LK_PD_LOAD_AUX
#define LK_PD_LOAD_AUX 0    /* cache policy of pyrDown's source loads (2 = nt: A/B in profiles/r03_*) */
#endif
struct PyrDownArgs {
    CPlanes src;
    Planes dst;
    int H, W, Hd, Wd;
    int gx, gy;        // tiles per plane
    int th;            // output rows per block
};

constexpr int PD_WCOLS = 62;                          // valid output columns per wave

// (bound_ctrl: a lane without a source -- lane 0 of a right shift, lane 63 of a left shift -- reads 0.  Rounds 1-3 passed a zero "old"
// operand to update_dpp instead, which costs a v_mov_b32 0 in front of every DPP move: 28 of ~400 vector instructions per marched row of the
// fused 3 x 3 kernel, whose vector pipes are 92 % busy.)
__device__ __forceinline__ double dpp_wave_shr1(double v)      // lane l <- lane l-1
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x138, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x138, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_wave_shl1(double v)      // lane l <- lane l+1
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x130, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x130, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(256) void pyrdown_kernel(PyrDownArgs a)
{
    const unsigned n = gridDim.x;
    unsigned tile = xcd_tile_id(blockIdx.x, n);
    const int bx = tile % a.gx; tile /= a.gx;
    const int by = tile % a.gy;
    const int plane = tile / a.gy;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int H = a.H, W = a.W;
    const int ox = (bx * 4 + wave) * PD_WCOLS + lane - 1;   // output column of this lane (halo: -1 / +62)
    const int oy0 = by * a.th, oy1 = min(oy0 + a.th, a.Hd);
    if ((bx * 4 + wave) * PD_WCOLS >= a.Wd) return;         // whole wave outside (wave-uniform)
    const double *s = a.src.at(plane);
    const int pitch = (int)a.src.pitch;
    const int c0 = 2 * ox;                                  // source columns c0, c0+1
    // wave-uniform: only the first / last wave of a row has lanes outside the image (a per-lane condition would
    // make the compiler select between two addresses and issue two 8-byte loads per row for every lane)
    const bool vec = __all(c0 >= 0 && c0 + 1 < W);
    const int r0c = r101(c0, W), r1c = r101(c0 + 1, W);
    // source rows through a buffer descriptor based at the strip's first row (scalar row offset, 4-byte
    // alignment suffices for the 16-byte access, 32-bit offsets for any plane size)
    const int row_lo = H < 4 ? 0 : max(2 * oy0 - 2, 0);
    const __amdgpu_buffer_rsrc_t rS = buf_rsrc(s + (long long)row_lo * pitch, ((long long)(H - 1 - row_lo) * pitch + W) * 8);
    const int pitchb = pitch * 8;

    // one source row: 16-byte load of this lane's column pair (issued early) ...
    auto ld = [&](int sy, double &cx, double &cy) __attribute__((always_inline)) {
        const int so = (r101(sy, H) - row_lo) * pitchb;     // row index is wave-uniform
        if (vec) {
            typedef unsigned int lk_u4 __attribute__((ext_vector_type(4)));
            const lk_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rS, c0 * 8, so, LK_PD_LOAD_AUX);
            cx = __hiloint2double((int)v.y, (int)v.x); cy = __hiloint2double((int)v.w, (int)v.z);
        } else {
            cx = buf_ld(rS, r0c * 8, so); cy = buf_ld(rS, r1c * 8, so);
        }
    };
    // ... and its horizontal result for this lane's output column (neighbour taps by DPP wave shifts)
    auto hf = [&](double cx, double cy) __attribute__((always_inline)) -> double {
        const double lx = dpp_wave_shr1(cx), ly = dpp_wave_shr1(cy), rx = dpp_wave_shl1(cx);
        return cx * 6 + (ly + cy) * 4 + lx + rx;
    };

    double ax, ay, bx2, by2, cx2, cy2;
    ld(2 * oy0 - 2, ax, ay); ld(2 * oy0 - 1, bx2, by2); ld(2 * oy0, cx2, cy2);
    double h0 = hf(ax, ay), h1 = hf(bx2, by2), h2 = hf(cx2, cy2);
    const bool out_lane = lane >= 1 && lane <= PD_WCOLS && ox < a.Wd;
    const int dpb = (int)a.dst.pitch * 8, ox8 = ox * 8;
    const __amdgpu_buffer_rsrc_t rD = buf_rsrc(a.dst.at(plane) + (long long)oy0 * a.dst.pitch,
                                               ((long long)(oy1 - 1 - oy0) * a.dst.pitch + a.Wd) * 8);
    int so = 0;                                             // byte offset of output row oy inside the strip
    // two output rows per iteration: four independent 16-byte loads in flight per lane
    int oy = oy0;
    for (; oy + 1 < oy1; oy += 2) {
        double p3x, p3y, p4x, p4y, p5x, p5y, p6x, p6y;
        ld(2 * oy + 1, p3x, p3y); ld(2 * oy + 2, p4x, p4y); ld(2 * oy + 3, p5x, p5y); ld(2 * oy + 4, p6x, p6y);
        const double h3 = hf(p3x, p3y), h4 = hf(p4x, p4y), h5 = hf(p5x, p5y), h6 = hf(p6x, p6y);
        const double o0 = (h2 * 6 + (h1 + h3) * 4 + h0 + h4) * (1.0 / 256);
        const double o1 = (h4 * 6 + (h3 + h5) * 4 + h2 + h6) * (1.0 / 256);
        if (out_lane) { buf_st_cached(rD, ox8, so, o0); buf_st_cached(rD, ox8, so + dpb, o1); }   // next level / LK re-read it soon
        so += 2 * dpb;
        h0 = h4; h1 = h5; h2 = h6;
    }
    if (oy < oy1) {
        double p3x, p3y, p4x, p4y;
        ld(2 * oy + 1, p3x, p3y); ld(2 * oy + 2, p4x, p4y);
        const double h3 = hf(p3x, p3y), h4 = hf(p4x, p4y);
        if (out_lane) buf_st_cached(rD, ox8, so, (h2 * 6 + (h1 + h3) * 4 + h0 + h4) * (1.0 / 256));
    }
}

// ------------------------------------------------------------------------------------------------
// Two pyramid levels in one pass (round 4): level 1 = cv::pyrDown(src) and level 2 = cv::pyrDown(level 1), the level-1 rows
// going to memory AND, still in registers, through the second filter -- the source is read once and level 1 is never read
// back (per 1080p frame: 16.6 + 4.1 + 1.0 MB moved instead of 16.6 + 4.1 + 4.1 + 1.0).  Same expressions as pyrdown_kernel at
// both levels (level 2 is computed from the very doubles that are stored as level 1): bit-identical to two passes.
// Lane l of wave column w holds level-1 column ox = 58 w + l - 3.  Lanes 1..62 compute level 1 (neighbour source pairs by DPP,
// as in pyrdown_kernel), lanes 3..60 store it; the level-2 column 2 x2 = ox sits in the ODD lanes 3..59 (29 per wave) and takes
// its five horizontal taps from lanes l - 2 .. l + 2 by DPP wave shifts: the four extra level-1 columns per wave are the halo
// of the second filter.  At the left / right image border the halo lanes take the REFLECTED level-1 column from the lane that
// holds it (reflect-101 of level 1 is not what pyrDown of the reflected source gives).  Vertically a block computes the
// level-1 rows oy0 - 2 .. oy1 of its strip (stores oy0 .. oy1 - 1) and keeps the last five second-filter rows in registers;
// the top and bottom image rows use the reflected rows the ring already holds.  Host-checked: W, H multiples of 4 (both levels
// halve exactly), W / 2 >= 8, H / 2 >= 4, strip height a multiple of 4.
// ------------------------------------------------------------------------------------------------
struct PyrDown2Args {
    CPlanes src;
    Planes d1, d2;
    int H, W, H1, W1, H2, W2;
    int gx, gy;        // tiles per plane
    int th;            // level-1 rows stored per block (multiple of 4)
};
constexpr int PD2_WCOLS = 58;                         // level-1 columns stored per wave (= 29 level-2 columns)

__global__ __launch_bounds__(256) void pyrdown2_kernel(PyrDown2Args a)
{
    const unsigned n = gridDim.x;
    unsigned tile = xcd_tile_id(blockIdx.x, n);
    const int bx = tile % a.gx; tile /= a.gx;
    const int by = tile % a.gy;
    const int plane = tile / a.gy;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wcol = bx * 4 + wave;
    const int H = a.H, W = a.W, H1 = a.H1, W1 = a.W1;
    if (wcol * PD2_WCOLS >= W1) return;                     // whole wave outside (wave-uniform)
    const int ox = wcol * PD2_WCOLS + lane - 3;             // level-1 column of this lane
    const int oy0 = by * a.th, oy1 = min(oy0 + a.th, H1);   // level-1 rows this block stores (oy0 % 4 == 0, oy1 even)
    const int rs = max(oy0 - 2, 0), re = min(oy1, H1 - 1);  // level-1 rows it computes (rs even)
    const double *s = a.src.at(plane);
    const int pitch = (int)a.src.pitch;
    const int c0 = 2 * ox;                                  // source columns c0, c0 + 1
    const bool vec = __all(c0 >= 0 && c0 + 1 < W);          // wave-uniform (see pyrdown_kernel)
    const int r0c = r101(c0, W), r1c = r101(c0 + 1, W);
    const int row_lo = max(2 * rs - 2, 0);
    const __amdgpu_buffer_rsrc_t rS = buf_rsrc(s + (long long)row_lo * pitch, ((long long)(H - 1 - row_lo) * pitch + W) * 8);
    const int pitchb = pitch * 8;
    auto ld = [&](int sy, double &cx, double &cy) __attribute__((always_inline)) {
        const int so = (r101(sy, H) - row_lo) * pitchb;     // row index is wave-uniform
        if (vec) {
            typedef unsigned int lk_u4 __attribute__((ext_vector_type(4)));
            const lk_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rS, c0 * 8, so, LK_PD_LOAD_AUX);
            cx = __hiloint2double((int)v.y, (int)v.x); cy = __hiloint2double((int)v.w, (int)v.z);
        } else {
            cx = buf_ld(rS, r0c * 8, so); cy = buf_ld(rS, r1c * 8, so);
        }
    };
    auto hf = [&](double cx, double cy) __attribute__((always_inline)) -> double {   // level-1 horizontal filter (pyrdown_kernel's)
        const double lx = dpp_wave_shr1(cx), ly = dpp_wave_shr1(cy), rx = dpp_wave_shl1(cx);
        return cx * 6 + (ly + cy) * 4 + lx + rx;
    };
    // reflect-101 of the LEVEL-1 columns for the second filter: a halo lane outside [0, W1) takes the value of the lane that
    // holds the reflected column (always in this wave: -1 -> 1, -2 -> 2, W1 -> W1 - 2, W1 + 1 -> W1 - 3)
    int fsrc = lane;
    if (ox < 0) fsrc = lane - 2 * ox;
    if (ox >= W1) fsrc = lane + 2 * (W1 - 1 - ox);
    fsrc = min(max(fsrc, 0), 63);
    const bool need_fix = __any(ox < 0 || ox >= W1);        // wave-uniform: first / last wave of a row
    auto hf2 = [&](double v) __attribute__((always_inline)) -> double {             // level-2 horizontal filter on level-1 lanes
        if (need_fix) v = __shfl(v, fsrc, 64);
        const double l1 = dpp_wave_shr1(v), r1 = dpp_wave_shl1(v);
        const double l2 = dpp_wave_shr1(l1), r2 = dpp_wave_shl1(r1);
        return v * 6 + (l1 + r1) * 4 + l2 + r2;
    };
    const bool st1_lane = lane >= 3 && lane <= 60 && ox < W1;
    const bool st2_lane = (lane & 1) && lane >= 3 && lane <= 59 && ox < W1;          // ox even: level-2 column ox / 2
    const int d1pb = (int)a.d1.pitch * 8, d2pb = (int)a.d2.pitch * 8;
    const __amdgpu_buffer_rsrc_t rD1 = buf_rsrc(a.d1.at(plane) + (long long)oy0 * a.d1.pitch, ((long long)(oy1 - 1 - oy0) * a.d1.pitch + W1) * 8);
    const int y20 = oy0 >> 1, y21 = oy1 >> 1;               // level-2 rows of this block
    const __amdgpu_buffer_rsrc_t rD2 = buf_rsrc(a.d2.at(plane) + (long long)y20 * a.d2.pitch, ((long long)(y21 - 1 - y20) * a.d2.pitch + a.W2) * 8);
    const int vo1 = st1_lane ? ox * 8 : (int)0x80000000, vo2 = st2_lane ? (ox >> 1) * 8 : (int)0x80000000;   // out of range: the store is dropped

    double gm4 = 0, gm3 = 0, gm2 = 0, gm1 = 0, g0 = 0;      // second-filter rows r - 4 .. r of the level-1 row r just made
    // one finished level-1 row r (wave-uniform): store it, run it through the second filter, emit the level-2 row it completes
    auto row_done = [&](int r, double v) __attribute__((always_inline)) {
        if (r >= oy0 && r < oy1) buf_st_cached(rD1, vo1, (r - oy0) * d1pb, v);      // (the LK kernels re-read it soon)
        gm4 = gm3; gm3 = gm2; gm2 = gm1; gm1 = g0; g0 = hf2(v);
        if (!(r & 1) && r >= oy0 + 2) {                     // level-2 row (r - 2) / 2: rows r - 4 .. r (top of the image: -2 -> 2, -1 -> 1)
            const double t4 = r == 2 ? g0 : gm4, t3 = r == 2 ? gm1 : gm3;
            buf_st_cached(rD2, vo2, (((r - 2) >> 1) - y20) * d2pb, (gm2 * 6 + (t3 + gm1) * 4 + t4 + g0) * (1.0 / 256));
        }
        if (r == H1 - 1)                                    // the last level-2 row: rows H1 - 4 .. H1 -> H1 - 2 (H1 is even)
            buf_st_cached(rD2, vo2, (((H1 >> 1) - 1) - y20) * d2pb, (gm1 * 6 + (gm2 + g0) * 4 + gm3 + gm1) * (1.0 / 256));
    };

    double ax, ay, bx2, by2, cx2, cy2;
    ld(2 * rs - 2, ax, ay); ld(2 * rs - 1, bx2, by2); ld(2 * rs, cx2, cy2);
    double h0 = hf(ax, ay), h1 = hf(bx2, by2), h2 = hf(cx2, cy2);
    int r = rs;
    for (; r + 1 <= re; r += 2) {                           // two level-1 rows per iteration: four 16-byte loads in flight per lane
        double p3x, p3y, p4x, p4y, p5x, p5y, p6x, p6y;
        ld(2 * r + 1, p3x, p3y); ld(2 * r + 2, p4x, p4y); ld(2 * r + 3, p5x, p5y); ld(2 * r + 4, p6x, p6y);
        const double h3 = hf(p3x, p3y), h4 = hf(p4x, p4y), h5 = hf(p5x, p5y), h6 = hf(p6x, p6y);
        const double o0 = (h2 * 6 + (h1 + h3) * 4 + h0 + h4) * (1.0 / 256);
        const double o1 = (h4 * 6 + (h3 + h5) * 4 + h2 + h6) * (1.0 / 256);
        row_done(r, o0);
        row_done(r + 1, o1);
        h0 = h4; h1 = h5; h2 = h6;
    }
    if (r <= re) {
        double p3x, p3y, p4x, p4y;
        ld(2 * r + 1, p3x, p3y); ld(2 * r + 2, p4x, p4y);
        const double h3 = hf(p3x, p3y), h4 = hf(p4x, p4y);
        row_done(r, (h2 * 6 + (h1 + h3) * 4 + h0 + h4) * (1.0 / 256));
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
//
// Row-marching: thread = one column x, lanes = consecutive columns (a gather instruction touches 4-5
// full 128-byte lines; the two horizontal taps of a row are one 16-byte load), a wave marches down a
// strip of `th` rows two at a time.  cv::pyrUp is evaluated incrementally: the three horizontally
// filtered coarse rows s(i-1), s(i), s(i+1) of the current row pair live in registers and ONE new
// coarse row (3 loads per flow plane) is fetched per pair (branch-free asymmetric borders, same
// roundings as pyrup_at()).  Block = 4 waves = 256 columns; XCD-aware 1-D grid.
// ------------------------------------------------------------------------------------------------
struct WarpArgs {
    CPlanes img1, img2, cu, cv;
    Planes out1, out2;
    int H, W;
    int gx, gy;
    int th;           // rows per block (even)
};

#ifndef LK_WARP_UNCOND
#define LK_WARP_UNCOND 1    /* warp_kernel: all eight gathers of a row pair unconditionally and up front (A/B: profiles/r03_e_*) */
#endif
#ifndef LK_WARP_WAVES
#define LK_WARP_WAVES 1
#endif
// M24: the launcher has checked that the image pitch and height are below 2^24 (24-bit multiplies in the tap addresses)
template <bool M24>
__global__ __launch_bounds__(256, LK_WARP_WAVES) void warp_kernel(WarpArgs a)
{
    const unsigned n = gridDim.x;
    unsigned tile = xcd_tile_id(blockIdx.x, n);
    const int bx = tile % a.gx; tile /= a.gx;
    const int by = tile % a.gy;
    const int p = tile / a.gy;
    const int x = bx * 256 + threadIdx.x;
    const int H = a.H, W = a.W, Hc = H >> 1, Wc = W >> 1;
    const int y0 = by * a.th, y1 = min(y0 + a.th, H);
    if (x >= W) return;

    // per-thread pyrUp column selectors (see pyrup_h()); tap positions as byte offsets
    const int j = x >> 1;
    const int jm8 = (j > 0 ? j - 1 : 0) * 8, jc8 = j * 8, jp8 = (j < Wc - 1 ? j + 1 : Wc - 1) * 8;
    const bool single = Wc == 1, left = j == 0, right = j == Wc - 1, odd = (x & 1) != 0;
    // (weights kept as floats -- exact small integers -- to stay within 64 VGPRs = 8 waves per SIMD)
    const float mBf = single ? 8.0f : (right ? 7.0f : 6.0f);
    const float wAf = (single || left) ? 0.0f : 1.0f;
    const float wCf = (single || right) ? 0.0f : (left ? 2.0f : 1.0f);
    const bool dup = single || right;
    // coarse flow rows and output rows through buffer descriptors based at the strip's first row: the
    // column byte offset is loop-invariant and the row offset wave-uniform (no vector address arithmetic)
    const int i0 = y0 >> 1;                                 // y0 is even
    const int crow_lo = max(i0 - 1, 0);
    const int cpub = (int)a.cu.pitch * 8, cpvb = (int)a.cv.pitch * 8;
    const __amdgpu_buffer_rsrc_t rCU = buf_rsrc(a.cu.at(p) + (long long)crow_lo * a.cu.pitch, ((long long)(Hc - 1 - crow_lo) * a.cu.pitch + Wc) * 8);
    const __amdgpu_buffer_rsrc_t rCV = buf_rsrc(a.cv.at(p) + (long long)crow_lo * a.cv.pitch, ((long long)(Hc - 1 - crow_lo) * a.cv.pitch + Wc) * 8);
    auto srow = [&](int k) { return (k < 0 ? min(1, Hc - 1) : (k > Hc - 1 ? Hc - 1 : k)) - crow_lo; };   // r101(2k,2Hc)/2
    auto htap = [&](__amdgpu_buffer_rsrc_t r, int so) __attribute__((always_inline)) -> double {
        const double A0 = buf_ld(r, jm8, so), B = buf_ld(r, jc8, so), C0 = buf_ld(r, jp8, so);
        // a tap with weight 0 is one cv::pyrUp does not read (image borders): dropped before the multiplication, so
        // that a non-finite flow value spreads exactly as far as in the reference
        const double A = wAf != 0.0f ? A0 : 0.0, C = wCf != 0.0f ? C0 : 0.0;
        const double ev = (A * (double)wAf + B * (double)mBf) + C * (double)wCf;
        const double od = (B + (dup ? B : C0)) * 4;
        return odd ? od : ev;
    };

    const double *I1 = a.img1.at(p), *I2 = a.img2.at(p);
    const int ip = (int)a.img1.pitch, op1b = (int)a.out1.pitch * 8, op2b = (int)a.out2.pitch * 8;
    const __amdgpu_buffer_rsrc_t rO1 = buf_rsrc(a.out1.at(p) + (long long)y0 * a.out1.pitch, ((long long)(y1 - 1 - y0) * a.out1.pitch + W) * 8);
    const __amdgpu_buffer_rsrc_t rO2 = buf_rsrc(a.out2.at(p) + (long long)y0 * a.out2.pitch, ((long long)(y1 - 1 - y0) * a.out2.pitch + W) * 8);
    const int x8 = x * 8;
    const float xf = (float)x;

    int i = i0;
    double hu0 = htap(rCU, srow(i - 1) * cpub), hu1 = htap(rCU, srow(i) * cpub);
    double hv0 = htap(rCV, srow(i - 1) * cpvb), hv1 = htap(rCV, srow(i) * cpvb);
    for (int y = y0; y < y1; y += 2, ++i) {
        const double hu2 = htap(rCU, srow(i + 1) * cpub), hv2 = htap(rCV, srow(i + 1) * cpvb);
        const double upu0 = (hu0 + hu1 * 6 + hu2) * (1.0 / 64), upu1 = ((hu1 + hu2) * 4) * (1.0 / 64);
        const double upv0 = (hv0 + hv1 * 6 + hv2) * (1.0 / 64), upv1 = ((hv1 + hv2) * 4) * (1.0 / 64);
        // map build (ParallelVideoPyr.cpp:674-678), float32 arithmetic
        const float hx0 = (float)upu0 / 2, hy0 = (float)upv0 / 2, hx1 = (float)upu1 / 2, hy1 = (float)upv1 / 2;
        const float yf0 = (float)y, yf1 = (float)(y + 1);
        const int so1 = (y - y0) * op1b, so2 = (y - y0) * op2b;
        // all four gathers in flight before the first store (H is even: row y+1 exists)
#if LK_WARP_UNCOND
        // all eight gathers of the row pair unconditionally and up front (lk_device.h remap_taps / remap_blend: bit-identical)
        const RemapTaps ta0 = remap_taps<M24>(ip, H, W, xf + hx0, yf0 + hy0), tb0 = remap_taps<M24>(ip, H, W, xf - hx0, yf0 - hy0);
        const RemapTaps ta1 = remap_taps<M24>(ip, H, W, xf + hx1, yf1 + hy1), tb1 = remap_taps<M24>(ip, H, W, xf - hx1, yf1 - hy1);
        const lk_d2u g0 = *reinterpret_cast<const lk_d2u *>(I1 + ta0.off), g1 = *reinterpret_cast<const lk_d2u *>(I1 + ta0.off + ip);
        const lk_d2u g2 = *reinterpret_cast<const lk_d2u *>(I2 + tb0.off), g3 = *reinterpret_cast<const lk_d2u *>(I2 + tb0.off + ip);
        const lk_d2u g4 = *reinterpret_cast<const lk_d2u *>(I1 + ta1.off), g5 = *reinterpret_cast<const lk_d2u *>(I1 + ta1.off + ip);
        const lk_d2u g6 = *reinterpret_cast<const lk_d2u *>(I2 + tb1.off), g7 = *reinterpret_cast<const lk_d2u *>(I2 + tb1.off + ip);
        const double a0 = remap_blend(ta0, g0, g1, H, W), b0 = remap_blend(tb0, g2, g3, H, W);
        const double a1 = remap_blend(ta1, g4, g5, H, W), b1 = remap_blend(tb1, g6, g7, H, W);
#else
        const double a0 = remap_bilinear(I1, ip, H, W, xf + hx0, yf0 + hy0);
        const double b0 = remap_bilinear(I2, ip, H, W, xf - hx0, yf0 - hy0);
        const double a1 = remap_bilinear(I1, ip, H, W, xf + hx1, yf1 + hy1);
        const double b1 = remap_bilinear(I2, ip, H, W, xf - hx1, yf1 - hy1);
#endif
        buf_st(rO1, x8, so1, a0); buf_st(rO2, x8, so2, b0);
        buf_st(rO1, x8, so1 + op1b, a1); buf_st(rO2, x8, so2 + op2b, b1);
        hu0 = hu1; hu1 = hu2; hv0 = hv1; hv1 = hv2;
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
    int gx, gy;       // lk_split: column blocks / row strips per pair (1-D grid, XCD-aware tile order)
    unsigned nmain;   // lk_split: marching workgroups; workgroups nmain .. gridDim.x - 1 do the frame pixels (0: separate launch)
    unsigned nbb;     // lk_split: frame workgroups per pair
    // lk_split, two strip heights in one launch (split_tile below): workgroups 0 .. tilesA - 1 march the first csA column strips
    // (a column strip = one column block of one pair) in gy strips of th rows, the others the remaining csB in gyB strips of thB
    // rows.  One height: tilesA = nmain, csB = 0.
    unsigned tilesA, csA, csB;
    int gyB, thB;
    // lk_split: image column where the OUTPUT columns of column block 0 start (even, <= R; 0 in every other kernel).  The frame of
    // width R belongs to the frame workgroups, so the marching blocks only need to cover columns R .. W - R - 1: starting block 0
    // at R & ~1 instead of 0 saves a whole column block when W - 2R fits one block fewer than W does (120 columns at 15 x 15: one
    // block of 110 instead of two; 1920 and 960 at 21 x 21: 18 / 9 instead of 19 / 10).  Same lanes' arithmetic for every
    // column (column parity is kept), so results do not depend on it.
    int xorg;
};

// workgroup -> (pair, column block, first row, rows per strip) of the wave-specialised kernel: XCD-aware within each phase
// (xcd_tile_id above; tilesA is a multiple of 8), column blocks fastest, then strips, then runs of gx column strips (= pairs,
// when a phase starts on a pair)
struct SplitTile { int pair, bx, y0, th; };
__device__ __forceinline__ SplitTile split_tile(const LKArgs &a, unsigned blk, unsigned nmain)
{
    const bool pb = blk >= a.tilesA;
    unsigned tile = pb ? xcd_tile_id(blk - a.tilesA, nmain - a.tilesA) : xcd_tile_id(blk, a.tilesA < nmain ? a.tilesA : nmain);
    const unsigned gyp = (unsigned)(pb ? a.gyB : a.gy), np = pb ? a.csB : a.csA, c0 = pb ? a.csA : 0u, gx = (unsigned)a.gx;
    const unsigned per = gx * gyp, run = tile / per, within = tile - run * per;
    const unsigned left = np - run * gx, rs = left < gx ? left : gx;         // (the last run of a phase may be short)
    const unsigned by = within / rs, cs = c0 + run * gx + (within - by * rs);
    SplitTile t;
    t.pair = (int)(cs / gx); t.bx = (int)(cs - (unsigned)t.pair * gx);
    t.th = pb ? a.thB : a.th; t.y0 = (int)by * t.th;
    return t;
}

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
// Edge map: the step of the video loop that PRODUCES the hot call's CV_64F inputs from the 8-bit frame
// difference (ParallelVideoPyr.cpp:803-813): threshold TOZERO -> Sobel x + Sobel y (3x3, reflect-101)
// -> dilate x2 -> erode x2 (3x3 rectangle; two iterations == one 5x5, as OpenCV 2.4 itself rewrites
// them; pixels outside the image never win).  Every intermediate is a small exact integer, so the
// tile pipeline works in int32 LDS planes and converts once at the end:
//   thresholded pixels (T+10)^2 -> E = sx+sy (T+8)^2 -> row max -> column max = D (T+4)^2
//   -> row min -> column min = out T^2.          Block = 256 threads = a 64 x 16 output tile.
// ------------------------------------------------------------------------------------------------
// cv::addWeighted(cur, 1, prev, beta, 0) for one 8-bit pixel (ParallelVideoPyr.cpp:800): float arithmetic,
// every product and sum rounded, round half to even, saturated.
__device__ __forceinline__ int frame_diff_px(unsigned char cur, unsigned char prev, float beta)
{
    const float ta = __fmul_rn((float)cur, 1.0f);
    const float tb = __fmul_rn((float)prev, beta);
    const int v = __float2int_rn(__fadd_rn(__fadd_rn(ta, tb), 0.0f));
    return min(max(v, 0), 255);
}

struct EdgeArgs {
    const unsigned char *src;   // n_planes x rows x cols, contiguous planes, row pitch = src_pitch bytes
    const unsigned char *prev;  // optional (same layout): the input becomes src - rate*prev (frame_diff_kernel's rule)
    float beta;                 // (float)(-rate)
    long long src_plane, src_pitch;
    Planes dst;
    int H, W, thresh;
    int gx, gy;
};

// Tile = 54 x 32 outputs: with the 5-pixel apron on each side the widest stage is exactly 64 columns,
// so every stage runs as "one wave per tile row, lane = column" (no index divisions, conflict-free LDS rows).
constexpr int EM_TW = 54, EM_TH = 32;

__global__ __launch_bounds__(256) void edge_map_kernel(EdgeArgs a)
{
    constexpr int PW = EM_TW + 10, PH = EM_TH + 10;     // thresholded pixels   (PW == 64)
    constexpr int EW = EM_TW + 8, EH = EM_TH + 8;       // Sobel sum
    constexpr int DW = EM_TW + 4, DH = EM_TH + 4;       // dilated
    static_assert(PW == 64, "stage rows are one wave wide");
    __shared__ int s_t[PH][64];
    __shared__ int s_e[EH][64];
    __shared__ int s_h[EH][64];                         // row max of E; later row min of D
    __shared__ int s_d[DH][64];

    const unsigned n = gridDim.x;
    unsigned tile = xcd_tile_id(blockIdx.x, n);
    const int bx = tile % a.gx; tile /= a.gx;
    const int by = tile % a.gy;
    const int plane = tile / a.gy;
    const int H = a.H, W = a.W;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x0 = bx * EM_TW, y0 = by * EM_TH;
    const unsigned char *src = a.src + (long long)plane * a.src_plane;

    // thresholded pixels, reflect-101 (only the Sobel taps of in-image E positions ever read them); all
    // byte loads of a thread are issued before the first LDS store (one memory latency per tile)
    {
        constexpr int NR = (PH + 3) / 4;
        int gx = x0 - 5 + lane;
        gx = gx < 0 ? -gx : gx;
        gx = gx >= W ? 2 * W - 2 - gx : gx;
        if (gx < 0 || gx >= W) gx = r101(x0 - 5 + lane, W);   // image narrower than the apron
        unsigned char v[NR], pv[NR];
        const unsigned char *prev = a.prev ? a.prev + (long long)plane * a.src_plane : nullptr;   // uniform
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const long long o = (long long)r101(y0 - 5 + min(wave + 4 * k, PH - 1), H) * a.src_pitch + gx;
            v[k] = src[o];
            pv[k] = prev ? prev[o] : (unsigned char)0;
        }
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            int d = v[k];
            if (prev) d = frame_diff_px(v[k], pv[k], a.beta);
            if (wave + 4 * k < PH) s_t[wave + 4 * k][lane] = d > a.thresh ? d : 0;
        }
    }
    __syncthreads();
    // E = Sobel x + Sobel y at (y0-4+r, x0-4+c); positions outside the image = "never wins" for the max
    for (int r = wave; r < EH; r += 4) {
        const int c = lane;
        const int yy = y0 - 4 + r, xx = x0 - 4 + c;
        int e = INT_MIN;
        if (c < EW && yy >= 0 && yy < H && xx >= 0 && xx < W) {
            // s_t index of (yy,xx) is [r+1][c+1]; s_t[r+1+dy][c+1+dx] holds pixel r101(yy+dy), r101(xx+dx),
            // i.e. the reflected neighbour cv::Sobel reads at the image border
            const int tl = s_t[r][c], tc = s_t[r][c + 1], tr = s_t[r][c + 2];
            const int ml = s_t[r + 1][c], mr = s_t[r + 1][c + 2];
            const int bl = s_t[r + 2][c], bc = s_t[r + 2][c + 1], br = s_t[r + 2][c + 2];
            const int sx = (tr - tl) + 2 * (mr - ml) + (br - bl);
            const int sy = (bl - tl) + 2 * (bc - tc) + (br - tr);
            e = sx + sy;
        }
        if (c < EW) s_e[r][c] = e;
    }
    __syncthreads();
    for (int r = wave; r < EH; r += 4) {                // row max over 5 columns
        const int c = lane;
        if (c < DW) {
            int m = s_e[r][c];
#pragma unroll
            for (int k = 1; k < 5; ++k) m = max(m, s_e[r][c + k]);
            s_h[r][c] = m;
        }
    }
    __syncthreads();
    for (int r = wave; r < DH; r += 4) {                // column max -> D at (y0-2+r, x0-2+c)
        const int c = lane;
        if (c < DW) {
            int m = s_h[r][c];
#pragma unroll
            for (int k = 1; k < 5; ++k) m = max(m, s_h[r + k][c]);
            const int yy = y0 - 2 + r, xx = x0 - 2 + c;
            s_d[r][c] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? m : INT_MAX;   // outside never wins the min
        }
    }
    __syncthreads();
    for (int r = wave; r < DH; r += 4) {                // row min over 5 columns (reuses s_h)
        const int c = lane;
        if (c < EM_TW) {
            int m = s_d[r][c];
#pragma unroll
            for (int k = 1; k < 5; ++k) m = min(m, s_d[r][c + k]);
            s_h[r][c] = m;
        }
    }
    __syncthreads();
    double *dst = a.dst.at(plane);
    for (int r = wave; r < EM_TH; r += 4) {             // column min -> output
        const int c = lane;
        const int yy = y0 + r, xx = x0 + c;
        if (c < EM_TW && yy < H && xx < W) {
            int m = s_h[r][c];
#pragma unroll
            for (int k = 1; k < 5; ++k) m = min(m, s_h[r + k][c]);
            dst[(long long)yy * a.dst.pitch + xx] = (double)m;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// f2: GaussianBlur(Size(k,k), sigma) on the 8UC3 frame + cvtColor(BGR2GRAY)  (ParallelVideoPyr.cpp:786-787)
// OpenCV 2.4 fixed-point path: integer kernel q (sum 256), row pass 8U -> 32S, column pass / 65536
// rounded half to even (the SSE2 column filter of the reference's x86-64 build), reflect-101;
// gray = (B*1868 + G*9617 + R*4899 + 8192) >> 14.  One block = 64 x 16 output pixels; the input tile
// (+ apron) and the row-pass result live in LDS.
// ------------------------------------------------------------------------------------------------
constexpr int BG_TW = 64, BG_TH = 32, BG_MAXR = 7;
struct BlurGrayArgs {
    const unsigned char *src;   // n_frames x rows x (cols*3) interleaved BGR
    long long src_plane, src_pitch;
    unsigned char *dst;         // n_frames x rows x cols
    long long dst_plane, dst_pitch;
    int H, W;
    int q[2 * BG_MAXR + 1];     // integer kernel, q[j] for tap j - R
    int gx, gy;
};

// R = ksize / 2.  Three phases per 64 x 32 tile: (0) the BGR tile + apron is de-interleaved into planar byte
// rows in LDS (reflect-101 folded into the gather); (1) row pass, 4 adjacent outputs per work item from
// (4 + 2R) bytes read as dwords; (2) column pass, 4 vertically adjacent outputs of all three channels per
// thread from (4 + 2R) LDS rows, rounding, BGR2GRAY, byte stores (64 consecutive bytes per wave and row).
template <int R>
__global__ __launch_bounds__(256) void blur_gray_kernel(BlurGrayArgs a)
{
    constexpr int IW = BG_TW + 2 * R, IH = BG_TH + 2 * R;
    constexpr int IWP = (IW + 3) / 4 * 4 + 4;           // padded planar byte row (dword multiple)
    constexpr int NB = 4 + 2 * R, ND = (NB + 3) / 4;     // bytes / dwords one row-pass item reads
    __shared__ __attribute__((aligned(16))) unsigned char s_in[3][IH][IWP];
    __shared__ __attribute__((aligned(16))) int s_row[3][IH][BG_TW];

    const unsigned n = gridDim.x;
    unsigned tile = xcd_tile_id(blockIdx.x, n);
    const int bx = tile % a.gx; tile /= a.gx;
    const int by = tile % a.gy;
    const int frame = tile / a.gy;
    const int H = a.H, W = a.W, t = threadIdx.x;
    const int x0 = bx * BG_TW, y0 = by * BG_TH;
    const unsigned char *src = a.src + (long long)frame * a.src_plane;

    // phase 0: gather + de-interleave; one wave per tile row (row index and row pointer wave-uniform),
    // lanes over the row's bytes; a single reflection covers every column unless the image is narrower
    // than the apron (then the general reflect-101 loop runs).  All byte loads of a thread are issued
    // before the first LDS store (one memory latency per tile instead of one per byte).
    {
        constexpr int NRI = (IH + 3) / 4, NBI = (IW * 3 + 63) / 64;
        const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
        int gofs[NBI], lofs[NBI];                            // per-lane global / LDS offsets inside a row
#pragma unroll
        for (int k = 0; k < NBI; ++k) {
            const int b = min(lane + 64 * k, IW * 3 - 1);
            const int px = b / 3, c = b - px * 3;
            int gx = x0 - R + px;
            gx = gx < 0 ? -gx : gx;
            gx = gx >= W ? 2 * W - 2 - gx : gx;
            if (gx < 0 || gx >= W) gx = r101(x0 - R + px, W);
            gofs[k] = gx * 3 + c;
            lofs[k] = c * (IH * IWP) + px;
        }
        unsigned char v[NRI][NBI];
#pragma unroll
        for (int it = 0; it < NRI; ++it) {
            const int rr = min(wave + 4 * it, IH - 1);
            const unsigned char *row = src + (long long)r101(y0 - R + rr, H) * a.src_pitch;
#pragma unroll
            for (int k = 0; k < NBI; ++k) v[it][k] = row[gofs[k]];
        }
#pragma unroll
        for (int it = 0; it < NRI; ++it) {
            const int rr = wave + 4 * it;
#pragma unroll
            for (int k = 0; k < NBI; ++k)
                if (rr < IH && lane + 64 * k < IW * 3) (&s_in[0][0][0])[lofs[k] + rr * IWP] = v[it][k];
        }
    }
    __syncthreads();
    for (int i = t; i < 3 * IH * (BG_TW / 4); i += 256) {    // phase 1: row pass, 8U -> 32S
        const int g = i % (BG_TW / 4), cr = i / (BG_TW / 4);
        const int rr = cr % IH, c = cr / IH;
        const unsigned *w = reinterpret_cast<const unsigned *>(&s_in[c][rr][4 * g]);
        unsigned d[ND];
#pragma unroll
        for (int k = 0; k < ND; ++k) d[k] = w[k];
        int o[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int v = (int)((d[k >> 2] >> (8 * (k & 3))) & 0xFFu);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k - e >= 0 && k - e <= 2 * R) o[e] += __mul24(a.q[k - e], v);   // 24-bit multiply: full rate
        }
        *reinterpret_cast<int4 *>(&s_row[c][rr][4 * g]) = make_int4(o[0], o[1], o[2], o[3]);
    }
    __syncthreads();
    for (int i = t; i < BG_TW * (BG_TH / 4); i += 256) {     // phase 2: column pass + BGR2GRAY
        const int xx = i % BG_TW, yg = i / BG_TW;
        const int x = x0 + xx;
        int px[4][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int acc[4] = {0, 0, 0, 0};                       // <= 255 * 65536: fits
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int v = s_row[c][4 * yg + k][xx];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (k - e >= 0 && k - e <= 2 * R) acc[e] += __mul24(a.q[k - e], v);   // both < 2^17
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int q = acc[e] >> 16;
                const int rem = acc[e] & 0xFFFF;
                if (rem > 0x8000 || (rem == 0x8000 && (q & 1))) ++q;      // half to even
                px[e][c] = min(max(q, 0), 255);
            }
        }
        if (x < W) {
            unsigned char *d = a.dst + (long long)frame * a.dst_plane + x;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int y = y0 + 4 * yg + e;
                if (y < H)
                    d[(long long)y * a.dst_pitch] =
                        (unsigned char)((px[e][0] * 1868 + px[e][1] * 9617 + px[e][2] * 4899 + 8192) >> 14);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// f2: diff = cur - rate * prev on 8UC1 frames (ParallelVideoPyr.cpp:800) = cv::addWeighted(cur, 1,
// prev, -rate, 0): FLOAT arithmetic, every product and sum rounded, round-half-even, saturate.
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// f2, ParallelVideoPyr.cpp:778: resize(frame, .., Size(dcols, drows), 0, 0, INTER_CUBIC) on 8UC3 frames, OpenCV 2.4's
// 8-bit path (imgwarp.cpp resizeGeneric_<HResizeCubic<uchar,int,short>, VResizeCubic<.., VResizeCubicVec_32s8u>>):
// 11-bit fixed-point weights from float cubic coefficients (A = -0.75), horizontal pass 8U -> 32S with clamped
// indices, vertical pass in FLOAT for the row elements the SSE2 vector op covers (all but the last width*3 mod 8)
// and in 32-bit integers (+ 2^21, >> 22) for the rest.  One thread per destination pixel; the weights of its column
// and row are recomputed per thread (the same IEEE operations in the same order as the host-side tables of OpenCV).
// ------------------------------------------------------------------------------------------------
struct ResizeArgs {
    const unsigned char *src; long long src_plane, src_pitch;
    unsigned char *dst; long long dst_plane, dst_pitch;
    int srows, scols, drows, dcols;
    double scale_x, scale_y;        // 1 / ((double)dsize / ssize)
};

__device__ __forceinline__ int cubic_weights_q11(int d, double scale, short w[4])
{
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    const int s = (int)floorf(f);
    f = __fsub_rn(f, (float)s);
    const float A = -0.75f, x = f, x1 = __fadd_rn(x, 1.0f), xm = __fsub_rn(1.0f, x);
    float c[4], t;
    t = __fmul_rn(A, x1); t = __fsub_rn(t, 5.0f * A); t = __fmul_rn(t, x1); t = __fadd_rn(t, 8.0f * A); t = __fmul_rn(t, x1); c[0] = __fsub_rn(t, 4.0f * A);
    t = __fmul_rn(A + 2.0f, x); t = __fsub_rn(t, A + 3.0f); t = __fmul_rn(t, x); t = __fmul_rn(t, x); c[1] = __fadd_rn(t, 1.0f);
    t = __fmul_rn(A + 2.0f, xm); t = __fsub_rn(t, A + 3.0f); t = __fmul_rn(t, xm); t = __fmul_rn(t, xm); c[2] = __fadd_rn(t, 1.0f);
    t = __fsub_rn(1.0f, c[0]); t = __fsub_rn(t, c[1]); c[3] = __fsub_rn(t, c[2]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int v = __float2int_rn(__fmul_rn(c[k], 2048.0f));      // saturate_cast<short>(w * INTER_RESIZE_COEF_SCALE)
        w[k] = (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
    }
    return s;
}

__global__ __launch_bounds__(256) void resize_cubic_kernel(ResizeArgs a)
{
    const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y, n = blockIdx.z;
    if (dx >= a.dcols) return;
    short al[4], be[4];
    const int xo = cubic_weights_q11(dx, a.scale_x, al), yo = cubic_weights_q11(dy, a.scale_y, be);
    int sx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sx[j] = min(max(xo - 1 + j, 0), a.scols - 1) * 3;
    const unsigned char *S = a.src + (long long)n * a.src_plane;
    int h[4][3];
    const bool inner = xo >= 1 && xo + 2 <= a.scols - 1;             // the four taps are 12 contiguous bytes
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned char *row = S + (long long)min(max(yo - 1 + k, 0), a.srows - 1) * a.src_pitch;
        unsigned char px[12];
        if (inner) {                                                 // three byte-aligned dword loads instead of twelve byte loads
            typedef unsigned int u32u __attribute__((aligned(1)));
            const u32u *q = reinterpret_cast<const u32u *>(row + sx[0]);
            const unsigned int w0 = q[0], w1 = q[1], w2 = q[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) { px[i] = (unsigned char)(w0 >> (8 * i)); px[4 + i] = (unsigned char)(w1 >> (8 * i)); px[8 + i] = (unsigned char)(w2 >> (8 * i)); }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) px[3 * j + c] = row[sx[j] + c];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int v = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) v += (int)px[3 * j + c] * (int)al[j];
            h[k][c] = v;
        }
    }
    const int vec_end = (a.dcols * 3) & ~7;                          // row elements the SSE2 vector op covers
    const float scale = 1.0f / (2048.0f * 2048.0f);
    const float b0 = __fmul_rn((float)be[0], scale), b1 = __fmul_rn((float)be[1], scale), b2 = __fmul_rn((float)be[2], scale),
                b3 = __fmul_rn((float)be[3], scale);
    unsigned char *o = a.dst + (long long)n * a.dst_plane + (long long)dy * a.dst_pitch + dx * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int v;
        if (dx * 3 + c < vec_end) {
            float s = __fmul_rn((float)h[0][c], b0);
            s = __fadd_rn(s, __fmul_rn((float)h[1][c], b1));
            s = __fadd_rn(s, __fmul_rn((float)h[2][c], b2));
            s = __fadd_rn(s, __fmul_rn((float)h[3][c], b3));
            v = __float2int_rn(s);
        } else {
            v = (h[0][c] * be[0] + h[1][c] * be[1] + h[2][c] * be[2] + h[3][c] * be[3] + (1 << 21)) >> 22;
        }
        o[c] = (unsigned char)min(max(v, 0), 255);
    }
}

struct FrameDiffArgs {
    const unsigned char *cur, *prev;
    long long src_plane, src_pitch;
    unsigned char *dst;
    long long dst_plane, dst_pitch;
    int H, W;
    float beta;                 // (float)(-rate)
};

__global__ __launch_bounds__(256) void frame_diff_kernel(FrameDiffArgs a)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, p = blockIdx.z;
    if (x >= a.W) return;
    const long long si = (long long)p * a.src_plane + (long long)y * a.src_pitch + x;
    a.dst[(long long)p * a.dst_plane + (long long)y * a.dst_pitch + x] =
        (unsigned char)frame_diff_px(a.cur[si], a.prev[si], a.beta);
}

// ------------------------------------------------------------------------------------------------
// convertTo(CV_64FC1, 1.0/255)  (SerialPyr.cpp:237)
// ------------------------------------------------------------------------------------------------
// A wave converts 512 consecutive bytes per iteration as four (2-byte load, 16-byte store) rounds, so that
// every load covers 128 contiguous bytes and every store 1 KiB contiguous; the unaligned head / tail
// (src odd or n not a multiple of 2) go through the scalar path.
__global__ __launch_bounds__(256) void u8_to_f64_kernel(const unsigned char *src, double *dst,
                                                        size_t n)
{
    const bool aligned = (reinterpret_cast<size_t>(src) & 1) == 0 && (reinterpret_cast<size_t>(dst) & 15) == 0;
    const size_t npair = aligned ? n / 2 : 0;
    const unsigned short *s2 = reinterpret_cast<const unsigned short *>(src);
    double2 *d2 = reinterpret_cast<double2 *>(dst);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < npair; i += 4 * stride) {
        unsigned short v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = s2[i + k * stride];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            d2[i + k * stride] = make_double2((double)(v[k] & 0xFF) * (1.0 / 255), (double)(v[k] >> 8) * (1.0 / 255));
    }
    for (; i < npair; i += stride) {
        const unsigned short v = s2[i];
        d2[i] = make_double2((double)(v & 0xFF) * (1.0 / 255), (double)(v >> 8) * (1.0 / 255));
    }
    for (size_t j = 2 * npair + (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride)
        dst[j] = (double)src[j] * (1.0 / 255);
}

// ------------------------------------------------------------------------------------------------
// The 2 x 2 solve + cv::divide of a6 (ParallelVideoPyr.cpp:470-472, 510-520) as a stage of its own: S = five planes of n window
// sums (Sxx, Syy, Sxy, Sxt, Syt).  plain = 0: lk_solve (what every LK kernel runs: both quotients from one reciprocal iteration),
// plain = 1: two compiler-expanded IEEE divisions -- the pair must agree bit for bit on every input class.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void solve_kernel(const double *S, size_t n, double *u, double *v, int plain)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double s[5] = {S[i], S[n + i], S[2 * n + i], S[3 * n + i], S[4 * n + i]};
        double a, b;
        if (plain) lk_solve_plain(s, a, b);
        else lk_solve(s, a, b);
        u[i] = a; v[i] = b;
    }
}

// ------------------------------------------------------------------------------------------------
// Flow consumer (ParallelVideoPyr.cpp:845-873): count / sum-of-x / sum-of-y of the pixels with
// |flow| >= thresh.  Sums are integers < 2^53, accumulated in fp64 -> exact in any order.
// out[3*pair + {0,1,2}] must be zeroed by the caller (the launcher does it).
// ------------------------------------------------------------------------------------------------
// The reference tests sqrt(u*u + v*v) < thresh; sqrt is monotonic and correctly rounded, so the launcher
// passes thresh2 = the smallest double whose square root is >= thresh and the kernel compares the sum of
// squares against it (same decisions bit for bit, no fp64 square root per pixel).
__global__ __launch_bounds__(256) void centroid_kernel(CPlanes U, CPlanes V, int H, int W,
                                                       double thresh2, double *out)
{
    const int pair = blockIdx.z;
    const double *u = U.at(pair), *v = V.at(pair);
    double c = 0, sx = 0, sy = 0;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x < W) {
        const double xd = (double)x;
        int y = blockIdx.y;
        for (; y + 3 * (int)gridDim.y < H; y += 4 * gridDim.y) {      // four independent rows in flight
            double xf[4], yf[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                xf[k] = u[(long long)(y + k * gridDim.y) * U.pitch + x];
                yf[k] = v[(long long)(y + k * gridDim.y) * V.pitch + x];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (!(xf[k] * xf[k] + yf[k] * yf[k] < thresh2)) { c += 1.0; sx += xd; sy += (double)(y + k * gridDim.y); }
        }
        for (; y < H; y += gridDim.y) {
            const double xf = u[(long long)y * U.pitch + x], yf = v[(long long)y * V.pitch + x];
            if (!(xf * xf + yf * yf < thresh2)) { c += 1.0; sx += xd; sy += (double)y; }
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
