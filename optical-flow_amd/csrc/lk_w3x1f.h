// lk_w3x1f.h -- warp + single-level LK FUSED for the reference's own 3 x 3 window, ONE column per lane, gathers one row ahead.
//
// Same arithmetic and expression order as lk_w3x2.h; the window sums in the reference's own order (lk_device.h sum9_step):
// every marched row is PRODUCED (incremental cv::pyrUp of the coarse flow, float map, fixed-point bilinear remap of both
// UN-warped level images, ParallelVideoPyr.cpp:646-703) instead of loaded, the 3 x 3 window is two DPP wave shifts per
// field, the vertical window three rows in registers.  What differs is the schedule: the two-column kernel needs ~250
// VGPRs (2 waves per SIMD) and waits for the gathers of a row right after issuing them -- one exposed memory round trip per
// marched row.  One column per lane halves every per-lane state, so that (a) three waves per SIMD fit and (b) the four
// gathers of row r + 1 can stay in flight (16 VGPRs) while the derivatives, window sums and solves of row r are computed:
// the round trip hides behind the arithmetic of the same wave.
// Lanes 0 and 1 are the left halo (lane 1's product needs lane 0's pixel), lane 63 the right halo: 61 output columns per
// single-wave workgroup.
#pragma once
#include "lk_w3x2.h"

namespace lk {

constexpr int W3X1_XO = 61;                     // output columns per wave: lanes 2..62

template <bool M24>      // the launcher has checked that the image pitch and height are below 2^24 (24-bit multiplies in the tap addresses)
__global__ __launch_bounds__(64, 3) void lk_w3x1f_kernel(LKArgs a)
{
    constexpr int WIN = 3, R = 1;
    const int t = threadIdx.x;
    const int pair = blockIdx.z;
    const int H = a.H, W = a.W, Hc = H >> 1, Wc = W >> 1;   // H, W even (a coarser level exists)
    const int xa = (int)blockIdx.x * W3X1_XO - 2 + t;
    const int y0 = blockIdx.y * a.th;
    const int y1 = min(y0 + a.th, H);
    const int ip = (int)a.i1.pitch;
    const bool col_in = xa >= 0 && xa < W;
    const bool all_col_in = __builtin_amdgcn_ballot_w64(!col_in) == 0;
    const int xc = min(max(xa, 0), W - 1);
    const bool at_x0 = xa == 0;                              // left tap of column 0 = reflected column 1 = the right neighbour's pixel
    const double *I1 = a.i1.at(pair), *I2 = a.i2.at(pair);  // UN-warped level images (gathered)
    const bool is_out = t >= 2 && t <= 62 && xa < W;
    const int nmarch = (y1 - y0) + 2 * R;
    const int yy0 = y0 - R;
    const float xf = (float)xc;

    Sum9 win9[5];                                            // partial window sums in the reference's order (lk_device.h sum9_step)
#pragma unroll
    for (int f = 0; f < 5; ++f) win9[f].a = win9[f].b = 0.0;

    // ---- coarse flow: descriptors, horizontal weights of the column ----
    const int cpub = (int)a.cu.pitch * 8, cpvb = (int)a.cv.pitch * 8;
    const int crow_lo = max((y0 >> 1) - 2, 0);               // the prologue rows y0-2, y0-1 reach one coarse row further up
    auto up_srow = [&](int k) { return k < 0 ? min(1, Hc - 1) : (k > Hc - 1 ? Hc - 1 : k); };
    const __amdgpu_buffer_rsrc_t rCU = buf_rsrc(a.cu.at(pair) + (long long)crow_lo * a.cu.pitch, ((long long)(Hc - 1 - crow_lo) * a.cu.pitch + Wc) * 8);
    const __amdgpu_buffer_rsrc_t rCV = buf_rsrc(a.cv.at(pair) + (long long)crow_lo * a.cv.pitch, ((long long)(Hc - 1 - crow_lo) * a.cv.pitch + Wc) * 8);
    const UpSel1 us = make_upsel1(xc, Wc);
    auto hrow = [&](int k, double &u, double &v) __attribute__((always_inline)) {   // s(k), both planes
        const int sr = up_srow(k) - crow_lo;
        u = pyrup_h1v(buf_ld(rCU, us.jm, sr * cpub), buf_ld(rCU, us.jc, sr * cpub), buf_ld(rCU, us.jp, sr * cpub), us);
        v = pyrup_h1v(buf_ld(rCV, us.jm, sr * cpvb), buf_ld(rCV, us.jc, sr * cpvb), buf_ld(rCV, us.jp, sr * cpvb), us);
    };
    // cv::pyrUp at an arbitrary row (prologue only): vertical combination of three freshly fetched coarse rows
    auto up_at = [&](int yw, double &uu, double &vv) __attribute__((always_inline)) {
        const int k = yw >> 1;
        double u0, v0, u1, v1, u2, v2;
        hrow(k - 1, u0, v0); hrow(k, u1, v1); hrow(k + 1, u2, v2);
        if (yw & 1) { uu = ((u1 + u2) * 4) * (1.0 / 64); vv = ((v1 + v2) * 4) * (1.0 / 64); }
        else { uu = (u0 + u1 * 6 + u2) * (1.0 / 64); vv = (v0 + v1 * 6 + v2) * (1.0 / 64); }
    };
    // (only the first workgroup of a row holds image column 0, whose left tap is the reflected column 1: everywhere else the
    // right shift and the select are skipped -- wave-uniform)
    const bool has_x0 = blockIdx.x == 0;
    auto left_tap = [&](double c) __attribute__((always_inline)) {
        const double l = dpp_wave_shr1(c);
        if (!has_x0) return l;
        const double r = dpp_wave_shl1(c);
        return at_x0 ? r : l;
    };
    // gathers of one produced row in flight: coordinates + the two tap pairs of both images
    struct Fly { RemapTaps t1, t2; lk_d2u a0, a1, b0, b1; };
    auto issue = [&](int yw, double uu, double vv, Fly &g) __attribute__((always_inline)) {   // warp_kernel's map build, float32
        const float yf = (float)yw;
        const float hx = (float)uu / 2, hy = (float)vv / 2;
        g.t1 = remap_taps<M24>(ip, H, W, xf + hx, yf + hy);
        g.t2 = remap_taps<M24>(ip, H, W, xf - hx, yf - hy);
        g.a0 = *reinterpret_cast<const lk_d2u *>(I1 + g.t1.off); g.a1 = *reinterpret_cast<const lk_d2u *>(I1 + g.t1.off + ip);
        g.b0 = *reinterpret_cast<const lk_d2u *>(I2 + g.t2.off); g.b1 = *reinterpret_cast<const lk_d2u *>(I2 + g.t2.off + ip);
    };
    auto land = [&](const Fly &g, double &w1, double &w2) __attribute__((always_inline)) {
        w1 = remap_blend(g.t1, g.a0, g.a1, H, W);
        w2 = remap_blend(g.t2, g.b0, g.b1, H, W);
    };

    // ---- prologue: warped rows yy0-1 (previous) and yy0 (current), reflected into the image like cv::filter2D does;
    //      the gathers of row y0 (the row iteration 0 lands) are issued ----
    double p1, p2, c1, c2;
    Fly fly;
    double fu, fv;                                           // up-sampled flow of the row in flight
    {
        double uu, vv;
        Fly g;
        const int rp = r101(yy0 - 1, H), rc = r101(yy0, H);
        up_at(rp, uu, vv); issue(rp, uu, vv, g); land(g, p1, p2);
        up_at(rc, uu, vv); issue(rc, uu, vv, g); land(g, c1, c2);
    }
    // carried coarse rows of the pyrUp state machine, which runs ONE row ahead of the landing row: s(i-1), s(i), s(i+1) for the
    // row it produces; coarse taps in flight (TAP3 scheme of lk_march3.h / lk_w3x2f.h, slot = (row - y0) mod 3)
    double hu[3], hv[3];
    {
        const int i0 = y0 >> 1;
#pragma unroll
        for (int s = 0; s < 3; ++s) hrow(i0 - 1 + s, hu[s], hv[s]);
    }
    double tq3[3][3];
    auto tap_row = [&](int y) { return max(up_srow(((y + 1) >> 1) + 1) - crow_lo, 0); };
    {   // taps of rows y0-1 (slot 2) and y0 (slot 0), as the steps for rows y0-2 and y0-1 would have fetched them
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int yq = y0 - 1 + d;
            const bool odd = (yq & 1) != 0;
            const __amdgpu_buffer_rsrc_t rs = odd ? rCV : rCU;
            const int so = tap_row(yq | 1) * (odd ? cpvb : cpub);
            const int sl = d == 0 ? 2 : 0;
            tq3[sl][0] = buf_ld(rs, us.jm, so); tq3[sl][1] = buf_ld(rs, us.jc, so); tq3[sl][2] = buf_ld(rs, us.jp, so);
        }
        tq3[1][0] = tq3[1][1] = tq3[1][2] = 0.0;
    }
    // one step of the state machine: cv::pyrUp of the coarse flow at row yp (static slot KS = (yp - y0) mod 3), lk_w3x2f.h's schedule
    auto upstep = [&](auto ksc, int yp, double &uu, double &vv) __attribute__((always_inline)) {
        constexpr int KS = decltype(ksc)::value;
        {   // three coarse taps per row, one row ahead of the pyrUp that consumes them
            const int yq = yp + 1;
            const bool odd = (yq & 1) != 0;
            const __amdgpu_buffer_rsrc_t rs = odd ? rCV : rCU;
            const int so = tap_row(yq | 1) * (odd ? cpvb : cpub);
            constexpr int ts = (KS + 1) % 3;
            tq3[ts][0] = buf_ld(rs, us.jm, so); tq3[ts][1] = buf_ld(rs, us.jc, so); tq3[ts][2] = buf_ld(rs, us.jp, so);
        }
        if ((yp & 1) == 0 && yp > y0) {
            const double nu = hu[0], nv = hv[0];
            hu[0] = hu[1]; hu[1] = hu[2]; hu[2] = nu;
            hv[0] = hv[1]; hv[1] = hv[2]; hv[2] = nv;
        }
        if (yp & 1) {
            uu = ((hu[1] + hu[2]) * 4) * (1.0 / 64); vv = ((hv[1] + hv[2]) * 4) * (1.0 / 64);
            // next coarse row (u taps came with the previous, even, row -- slot KS+2 --, v taps with this one -- slot KS)
            hu[0] = pyrup_h1v(tq3[(KS + 2) % 3][0], tq3[(KS + 2) % 3][1], tq3[(KS + 2) % 3][2], us);
            hv[0] = pyrup_h1v(tq3[KS % 3][0], tq3[KS % 3][1], tq3[KS % 3][2], us);
        } else {
            uu = (hu[0] + hu[1] * 6 + hu[2]) * (1.0 / 64); vv = (hv[0] + hv[1] * 6 + hv[2]) * (1.0 / 64);
        }
    };
    upstep(std::integral_constant<int, 0>{}, y0, fu, fv);
    issue(min(y0, H - 1), fu, fv, fly);
    double lp1 = left_tap(p1), lp2 = left_tap(p2), lc1 = left_tap(c1), lc2 = left_tap(c2);
    // up-sampled flow of the produced rows (delay line): d1* = row yw-1, d2* = row yw-2
    double d1u = 0, d1v = 0, d2u = 0, d2v = 0;

    const int upb = (int)a.u.pitch * 8, vpb = (int)a.v.pitch * 8;
    const __amdgpu_buffer_rsrc_t rU = buf_rsrc(a.u.at(pair) + (long long)y0 * a.u.pitch, ((long long)(y1 - 1 - y0) * a.u.pitch + W) * 8);
    const __amdgpu_buffer_rsrc_t rV = buf_rsrc(a.v.at(pair) + (long long)y0 * a.v.pitch, ((long long)(y1 - 1 - y0) * a.v.pitch + W) * 8);
    const int vxo = xa * 8;

    auto row = [&](auto kc, int i) __attribute__((always_inline)) {
        constexpr int K = decltype(kc)::value;
        const int yy = yy0 + i;                                 // row whose products are formed now (current row c*)
        const int yw = yy + 1;                                  // row landing now (becomes c* at the rotate) = y0 + i
        const bool row_in = yy >= 0 && yy < H;
        const int y = yy - R;                                   // output row of this iteration = yw - 2
        const bool out_row = i >= 2 * R;
        // land the row issued one iteration ago, then put the next one in flight: it has this iteration's arithmetic to arrive
        double n1, n2;
        land(fly, n1, n2);
        const double uu = fu, vv = fv;                          // flow of row yw
        upstep(std::integral_constant<int, (K + 1) % 3>{}, yw + 1, fu, fv);
        issue(min(yw + 1, H - 1), fu, fv, fly);                 // (rows below the image are never used: their products are zeroed)

        double p[5];
        {
            const bool keep = row_in && col_in;
            double fx = f2d_x(lp1, p1, lc1, c1) + f2d_x(lp2, p2, lc2, c2);
            double fy = f2d_y(lp1, p1, lc1, c1) + f2d_y(lp2, p2, lc2, c2);
            const double ft = f2d_neg(lp1, p1, lc1, c1) + f2d_pos(lp2, p2, lc2, c2);
            if (!(row_in && all_col_in)) { fx = keep ? fx : 0.0; fy = keep ? fy : 0.0; }   // (wave-uniform test: rows / columns outside the image contribute 0)
            p[0] = fx * fx; p[1] = fy * fy; p[2] = fx * fy; p[3] = fx * ft; p[4] = fy * ft;
        }
        double S[5];
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            // get_Sum9's own order: nine sequential additions, row-major (the window of output row yy - 1 ends with this row)
            S[f] = sum9_step(win9[f], dpp_wave_shr1(p[f]), p[f], dpp_wave_shl1(p[f]));
        }
        p1 = c1; p2 = c2; lp1 = lc1; lp2 = lc2;
        c1 = n1; c2 = n2;
        lc1 = left_tap(c1); lc2 = left_tap(c2);

        if (out_row && is_out) {                                // flow of row y = yw - 2: its pyrUp term sits in d2*
            double u = 0.0, v = 0.0;
            if (y >= 1 && y <= H - 2 && xa >= 1 && xa <= W - 2) lk_solve_plain(S, u, v);   // (two plain divisions: the shared-reciprocal form needs 4 more registers than this kernel has at 3 waves per SIMD; same bits)
            buf_st(rU, vxo, (y - y0) * upb, u + d2u);
            buf_st(rV, vxo, (y - y0) * vpb, v + d2v);
        }
        d2u = d1u; d2v = d1v;
        d1u = uu; d1v = vv;
    };

    for (int base = 0; base < nmarch; base += WIN) {
        [&]<int... Ks>(std::integer_sequence<int, Ks...>) __attribute__((always_inline)) {
            ((base + Ks < nmarch ? (row(std::integral_constant<int, Ks>{}, base + Ks), 0) : 0), ...);
        }(std::make_integer_sequence<int, WIN>{});
    }
}

}  // namespace lk
