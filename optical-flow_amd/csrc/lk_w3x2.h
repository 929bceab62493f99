// lk_w3x2.h -- single-level dense LK for the reference's own 3 x 3 window, TWO adjacent columns per lane.
//
// At 3 x 3 a marched row is so short that the one-column kernel (lk_march3.h) is bound by the number of vector
// memory instructions it issues (8-byte-per-lane accesses cost the texture path as much as 16-byte ones).  Here a
// lane owns the column pair (xa, xa+1), xa even, of a single-wave block:
//   * one 16-byte load per image and row (the pair), three rows in flight; the left derivative tap of xa is the left
//     lane's second pixel (DPP wave shift; at image column 0 it is the reflected column 1 = the lane's own second
//     pixel), so there are no tap loads at all;
//   * the 3-tap horizontal window of both columns comes from the lane's own two products plus one DPP shift each way
//     (no LDS, no barrier); the nine terms of a window are added in the reference's own row-major order
//     (lk_device.h sum9_step), as in the one-column kernels: bit-identical results;
//   * both columns share one coarse column j = xa / 2 of the "+= pyrUp" term: three coarse taps per row serve both
//     (u plane on even output rows, v plane on odd ones -- the TAP3 scheme of lk_march3.h);
//   * flows leave as 16-byte stores.
// 3.5 vector memory instructions per pixel-row pair instead of 9 per pixel-row.  Needs an even image width (every
// pyramid level except possibly the coarsest); otherwise the launcher takes the one-column kernel.
#pragma once
#include "lk_march3.h"

namespace lk {

typedef unsigned int lk_u4 __attribute__((ext_vector_type(4)));

#ifndef LK_IMG_LOAD_AUX
#define LK_IMG_LOAD_AUX 0   /* cache policy of the LK kernels' 16-byte image loads (2 = nt: A/B in profiles/r03_*) */
#endif
__device__ __forceinline__ void buf_ld2(__amdgpu_buffer_rsrc_t r, int voff, int soff, double &x, double &y)
{
    const lk_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, LK_IMG_LOAD_AUX);
    x = __hiloint2double((int)v.y, (int)v.x);
    y = __hiloint2double((int)v.w, (int)v.z);
}
__device__ __forceinline__ void buf_st2(__amdgpu_buffer_rsrc_t r, int voff, int soff, double x, double y)
{
    lk_u4 v;
    v.x = (unsigned)__double2loint(x); v.y = (unsigned)__double2hiint(x);
    v.z = (unsigned)__double2loint(y); v.w = (unsigned)__double2hiint(y);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 2);      // streaming store, like buf_st()
}

constexpr int W3X2_XO = 124;                    // output columns per wave: lanes 1..62 x 2

template <bool HAS_UP>
__global__ __launch_bounds__(64, 2) void lk_w3x2_kernel(LKArgs a)
{
    constexpr int WIN = 3, R = 1, PF = 3;
    const int t = threadIdx.x;
    const int pair = blockIdx.z;
    const int H = a.H, W = a.W;                 // W even, >= 2 (host-checked): a column pair is all in or all out
    const int xa = (int)blockIdx.x * W3X2_XO - 2 + 2 * t;
    const int y0 = blockIdx.y * a.th;
    const int y1 = min(y0 + a.th, H);
    const int pitch = (int)a.i1.pitch;
    const bool col_in = xa >= 0 && xa < W;
    const int xc = min(max(xa, 0), W - 2);
    const bool at_x0 = xa == 0;                 // left tap of column 0 = reflected column 1
    const int row_lo = (H <= R + PF) ? 0 : max(y0 - R - PF, 0);
    const long long img_bytes = ((long long)(H - 1 - row_lo) * pitch + W) * 8;
    const __amdgpu_buffer_rsrc_t rI1 = buf_rsrc(a.i1.at(pair) + (long long)row_lo * pitch, img_bytes);
    const __amdgpu_buffer_rsrc_t rI2 = buf_rsrc(a.i2.at(pair) + (long long)row_lo * pitch, img_bytes);
    const int vx = xc * 8, pitchb = pitch * 8;
    const bool is_out = t >= 1 && t <= 62 && xa < W;

    const int nmarch = (y1 - y0) + 2 * R;
    const int yy0 = y0 - R;

    Sum9 w9A[5], w9B[5];                       // partial window sums of the two columns, reference order (lk_device.h sum9_step)
#pragma unroll
    for (int f = 0; f < 5; ++f) w9A[f].a = w9A[f].b = w9B[f].a = w9B[f].b = 0.0;

    auto img_row_off = [&](int yy) {
        int rr = yy < 0 ? -yy : yy;
        rr = rr >= H ? 2 * H - 2 - rr : rr;
        if (H <= R + PF) rr = r101(yy, H);
        return (rr - row_lo) * pitchb;
    };
    // left derivative tap of column xa for a row whose pair is (va, vb) in this lane
    auto left_tap = [&](double vb) __attribute__((always_inline)) {
        const double l = dpp_wave_shr1(vb);
        return at_x0 ? vb : l;
    };
    // previous row p*, current row c* ([a] = column xa, [b] = xa + 1), their left taps l*, and the rows in flight
    double p1a, p1b, p2a, p2b, c1a, c1b, c2a, c2b;
    buf_ld2(rI1, vx, (r101(yy0 - 1, H) - row_lo) * pitchb, p1a, p1b);
    buf_ld2(rI2, vx, (r101(yy0 - 1, H) - row_lo) * pitchb, p2a, p2b);
    buf_ld2(rI1, vx, (r101(yy0, H) - row_lo) * pitchb, c1a, c1b);
    buf_ld2(rI2, vx, (r101(yy0, H) - row_lo) * pitchb, c2a, c2b);
    double q1a[PF], q1b[PF], q2a[PF], q2b[PF];
#pragma unroll
    for (int r = 1; r < PF; ++r) {
        const int o = img_row_off(yy0 + r);
        buf_ld2(rI1, vx, o, q1a[r], q1b[r]);
        buf_ld2(rI2, vx, o, q2a[r], q2b[r]);
    }
    q1a[0] = q1b[0] = q2a[0] = q2b[0] = 0.0;
    double lp1 = left_tap(p1b), lp2 = left_tap(p2b), lc1 = left_tap(c1b), lc2 = left_tap(c2b);

    [[maybe_unused]] UpSel1 usA{}, usB{};
    [[maybe_unused]] double huA[3] = {0, 0, 0}, hvA[3] = {0, 0, 0}, huB[3] = {0, 0, 0}, hvB[3] = {0, 0, 0};
    [[maybe_unused]] const int Hc = H >> 1;
    [[maybe_unused]] const int cpub = (int)a.cu.pitch * 8, cpvb = (int)a.cv.pitch * 8;
    [[maybe_unused]] const int crow_lo = max((y0 >> 1) - 1, 0);
    auto up_srow = [&](int k) { return k < 0 ? min(1, Hc - 1) : (k > Hc - 1 ? Hc - 1 : k); };
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rCU =
        HAS_UP ? buf_rsrc(a.cu.at(pair) + (long long)crow_lo * a.cu.pitch, ((long long)(Hc - 1 - crow_lo) * a.cu.pitch + (W >> 1)) * 8) : rI1;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rCV =
        HAS_UP ? buf_rsrc(a.cv.at(pair) + (long long)crow_lo * a.cv.pitch, ((long long)(Hc - 1 - crow_lo) * a.cv.pitch + (W >> 1)) * 8) : rI2;
    if (HAS_UP) {
        usA = make_upsel1(xc, W >> 1);
        usB = make_upsel1(xc + 1, W >> 1);       // same coarse column j, odd-column weights
        const int i0 = y0 >> 1;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int sr = up_srow(i0 - 1 + s) - crow_lo;
            const double uA = buf_ld(rCU, usA.jm, sr * cpub), uB = buf_ld(rCU, usA.jc, sr * cpub), uC = buf_ld(rCU, usA.jp, sr * cpub);
            const double vA = buf_ld(rCV, usA.jm, sr * cpvb), vB = buf_ld(rCV, usA.jc, sr * cpvb), vC = buf_ld(rCV, usA.jp, sr * cpvb);
            huA[s] = pyrup_h1v(uA, uB, uC, usA); huB[s] = pyrup_h1v(uA, uB, uC, usB);
            hvA[s] = pyrup_h1v(vA, vB, vC, usA); hvB[s] = pyrup_h1v(vA, vB, vC, usB);
        }
    }
    [[maybe_unused]] double tq3[3][3];
    auto tap_row = [&](int y) { return max(up_srow(((y + 1) >> 1) + 1) - crow_lo, 0); };
    const int upb = (int)a.u.pitch * 8, vpb = (int)a.v.pitch * 8;
    const __amdgpu_buffer_rsrc_t rU = buf_rsrc(a.u.at(pair) + (long long)y0 * a.u.pitch, ((long long)(y1 - 1 - y0) * a.u.pitch + W) * 8);
    const __amdgpu_buffer_rsrc_t rV = buf_rsrc(a.v.at(pair) + (long long)y0 * a.v.pitch, ((long long)(y1 - 1 - y0) * a.v.pitch + W) * 8);
    const int vxo = xa * 8;

    auto row = [&](auto kc, int i) __attribute__((always_inline)) {
        constexpr int K = decltype(kc)::value;
        const int yy = yy0 + i;
        const bool row_in = yy >= 0 && yy < H;
        {   // image row yy + PF -> the queue slot freed by the previous rotate
            const int rn = img_row_off(yy + PF);
            buf_ld2(rI1, vx, rn, q1a[K % PF], q1b[K % PF]);
            buf_ld2(rI2, vx, rn, q2a[K % PF], q2b[K % PF]);
        }
        const int y = yy - R;
        const bool out_row = i >= 2 * R;
        if constexpr (HAS_UP) {                                 // three coarse taps per row (see lk_march3.h, TAP3), lookahead 1
            const int yq = y + 1;
            const bool odd = (yq & 1) != 0;
            const __amdgpu_buffer_rsrc_t rs = odd ? rCV : rCU;
            const int so = tap_row(yq | 1) * (odd ? cpvb : cpub);
            constexpr int ts = (K + 1) % 3;
            tq3[ts][0] = buf_ld(rs, usA.jm, so); tq3[ts][1] = buf_ld(rs, usA.jc, so); tq3[ts][2] = buf_ld(rs, usA.jp, so);
        }
        double pA[5], pB[5];
        {
            const bool keep = row_in && col_in;
            double fx = f2d_x(lp1, p1a, lc1, c1a) + f2d_x(lp2, p2a, lc2, c2a);
            double fy = f2d_y(lp1, p1a, lc1, c1a) + f2d_y(lp2, p2a, lc2, c2a);
            double ft = f2d_neg(lp1, p1a, lc1, c1a) + f2d_pos(lp2, p2a, lc2, c2a);
            fx = keep ? fx : 0.0; fy = keep ? fy : 0.0;
            pA[0] = fx * fx; pA[1] = fy * fy; pA[2] = fx * fy; pA[3] = fx * ft; pA[4] = fy * ft;
            fx = f2d_x(p1a, p1b, c1a, c1b) + f2d_x(p2a, p2b, c2a, c2b);
            fy = f2d_y(p1a, p1b, c1a, c1b) + f2d_y(p2a, p2b, c2a, c2b);
            ft = f2d_neg(p1a, p1b, c1a, c1b) + f2d_pos(p2a, p2b, c2a, c2b);
            fx = keep ? fx : 0.0; fy = keep ? fy : 0.0;
            pB[0] = fx * fx; pB[1] = fy * fy; pB[2] = fx * fy; pB[3] = fx * ft; pB[4] = fy * ft;
        }
        double SA[5], SB[5];
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            // get_Sum9's own order (nine sequential additions, row-major): independent of where the strip starts, so a
            // sharded / differently batched run reproduces the same bits
            SA[f] = sum9_step(w9A[f], dpp_wave_shr1(pB[f]), pA[f], pB[f]);
            SB[f] = sum9_step(w9B[f], pA[f], pB[f], dpp_wave_shl1(pA[f]));
        }
        // rotate: the only wait for the row prefetched PF iterations ago
        p1a = c1a; p1b = c1b; p2a = c2a; p2b = c2b; lp1 = lc1; lp2 = lc2;
        c1a = q1a[(K + 1) % PF]; c1b = q1b[(K + 1) % PF]; c2a = q2a[(K + 1) % PF]; c2b = q2b[(K + 1) % PF];
        lc1 = left_tap(c1b); lc2 = left_tap(c2b);

        if (out_row && is_out) {
            double uA = 0.0, vA = 0.0, uB = 0.0, vB = 0.0;
            const bool inner_row = y >= 1 && y <= H - 2;
            if (inner_row && xa >= 1) lk_solve(SA, uA, vA);             // xa <= W - 2 always (pair inside the image)
            if (inner_row && xa + 1 <= W - 2) lk_solve(SB, uB, vB);     // (a 3 x 3 window of a ring-interior pixel is never clipped)
            if constexpr (HAS_UP) {
                if ((y & 1) == 0 && y > y0) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        double *hu = c ? huB : huA, *hv = c ? hvB : hvA;
                        const double nu = hu[0], nv = hv[0];
                        hu[0] = hu[1]; hu[1] = hu[2]; hu[2] = nu;
                        hv[0] = hv[1]; hv[1] = hv[2]; hv[2] = nv;
                    }
                }
                if (y & 1) {
                    uA += ((huA[1] + huA[2]) * 4) * (1.0 / 64); vA += ((hvA[1] + hvA[2]) * 4) * (1.0 / 64);
                    uB += ((huB[1] + huB[2]) * 4) * (1.0 / 64); vB += ((hvB[1] + hvB[2]) * 4) * (1.0 / 64);
                    // next coarse row (u taps came with the previous, even, row; v taps with this one)
                    huA[0] = pyrup_h1v(tq3[(K + 2) % 3][0], tq3[(K + 2) % 3][1], tq3[(K + 2) % 3][2], usA);
                    huB[0] = pyrup_h1v(tq3[(K + 2) % 3][0], tq3[(K + 2) % 3][1], tq3[(K + 2) % 3][2], usB);
                    hvA[0] = pyrup_h1v(tq3[K % 3][0], tq3[K % 3][1], tq3[K % 3][2], usA);
                    hvB[0] = pyrup_h1v(tq3[K % 3][0], tq3[K % 3][1], tq3[K % 3][2], usB);
                } else {
                    uA += (huA[0] + huA[1] * 6 + huA[2]) * (1.0 / 64); vA += (hvA[0] + hvA[1] * 6 + hvA[2]) * (1.0 / 64);
                    uB += (huB[0] + huB[1] * 6 + huB[2]) * (1.0 / 64); vB += (hvB[0] + hvB[1] * 6 + hvB[2]) * (1.0 / 64);
                }
            }
            buf_st2(rU, vxo, (y - y0) * upb, uA, uB);
            buf_st2(rV, vxo, (y - y0) * vpb, vA, vB);
        }
    };

    for (int base = 0; base < nmarch; base += WIN) {
        [&]<int... Ks>(std::integer_sequence<int, Ks...>) __attribute__((always_inline)) {
            ((base + Ks < nmarch ? (row(std::integral_constant<int, Ks>{}, base + Ks), 0) : 0), ...);
        }(std::make_integer_sequence<int, WIN>{});
    }
}

}  // namespace lk
