// lk_w3x2f.h -- warp + single-level LK FUSED, for the reference's own 3 x 3 window (two columns per lane).
//
// At 3 x 3 every kernel of a level is HBM-bound, and the warp kernel writes two planes only for the LK kernel to
// read them straight back (32 of the level's 72 bytes per pixel).  This kernel marches like lk_w3x2_kernel but
// PRODUCES each image row instead of loading it: cv::pyrUp of the coarse flow at the row (incremental, one new
// coarse row per two rows), float map, fixed-point bilinear remap of both images -- the arithmetic of
// warp_kernel / remap_bilinear(), bit for bit -- and keeps the row in registers.  The up-sampled flow of the row
// is computed once: the warp uses it at once, the "+= pyrUp" term of the flow two marched rows later (a two-row
// delay line).  A 3 x 3 window needs one apron row and one halo lane per side, so the recomputed warp costs
// ~5 % extra work (it would be 25 % at 15 x 15, where LK is instruction-bound anyway).
// Used when the level has a coarser level (HAS_UP), an even width, and the caller does not ask for the warped
// images (LKFLOW_MUTATE_INPUTS at level 0); results are identical to warp_kernel + lk_w3x2_kernel.
#pragma once
#include "lk_w3x2.h"

#ifndef LK_W3F_UNCOND
#define LK_W3F_UNCOND 2     /* fused 3 x 3 kernel: 0 = remap_bilinear (conditional gathers), 1 = unconditional, two at a time, 2 = all eight of a row at once (A/B: profiles/r03_e_*) */
#endif

namespace lk {

__global__ __launch_bounds__(64, 2) void lk_w3x2f_kernel(LKArgs a)
{
    constexpr int WIN = 3, R = 1;
    const int t = threadIdx.x;
    const int pair = blockIdx.z;
    const int H = a.H, W = a.W, Hc = H >> 1, Wc = W >> 1;   // H, W even (a coarser level exists)
    const int xa = (int)blockIdx.x * W3X2_XO - 2 + 2 * t;
    const int y0 = blockIdx.y * a.th;
    const int y1 = min(y0 + a.th, H);
    const int ip = (int)a.i1.pitch;
    const bool col_in = xa >= 0 && xa < W;
    const int xc = min(max(xa, 0), W - 2);
    const bool at_x0 = xa == 0;
    const double *I1 = a.i1.at(pair), *I2 = a.i2.at(pair);  // UN-warped level images (gathered)
    const bool is_out = t >= 1 && t <= 62 && xa < W;
    const int nmarch = (y1 - y0) + 2 * R;
    const int yy0 = y0 - R;
    const float xfa = (float)xc, xfb = (float)(xc + 1);

    Sum9 w9A[5], w9B[5];                       // partial window sums of the two columns, reference order (lk_device.h sum9_step)
#pragma unroll
    for (int f = 0; f < 5; ++f) w9A[f].a = w9A[f].b = w9B[f].a = w9B[f].b = 0.0;

    // ---- coarse flow: descriptors, per-column horizontal weights, carried rows s(i-1), s(i), s(i+1) of row yw ----
    const int cpub = (int)a.cu.pitch * 8, cpvb = (int)a.cv.pitch * 8;
    const int crow_lo = max((y0 >> 1) - 2, 0);               // the prologue rows y0-2, y0-1 reach one coarse row further up
    auto up_srow = [&](int k) { return k < 0 ? min(1, Hc - 1) : (k > Hc - 1 ? Hc - 1 : k); };
    const __amdgpu_buffer_rsrc_t rCU = buf_rsrc(a.cu.at(pair) + (long long)crow_lo * a.cu.pitch, ((long long)(Hc - 1 - crow_lo) * a.cu.pitch + Wc) * 8);
    const __amdgpu_buffer_rsrc_t rCV = buf_rsrc(a.cv.at(pair) + (long long)crow_lo * a.cv.pitch, ((long long)(Hc - 1 - crow_lo) * a.cv.pitch + Wc) * 8);
    const UpSel1 usA = make_upsel1(xc, Wc), usB = make_upsel1(xc + 1, Wc);
    auto hrow = [&](int k, double &uA, double &uB, double &vA, double &vB) __attribute__((always_inline)) {   // s(k), both planes, both columns
        const int sr = up_srow(k) - crow_lo;
        const double a0 = buf_ld(rCU, usA.jm, sr * cpub), a1 = buf_ld(rCU, usA.jc, sr * cpub), a2 = buf_ld(rCU, usA.jp, sr * cpub);
        const double b0 = buf_ld(rCV, usA.jm, sr * cpvb), b1 = buf_ld(rCV, usA.jc, sr * cpvb), b2 = buf_ld(rCV, usA.jp, sr * cpvb);
        uA = pyrup_h1v(a0, a1, a2, usA); uB = pyrup_h1v(a0, a1, a2, usB);
        vA = pyrup_h1v(b0, b1, b2, usA); vB = pyrup_h1v(b0, b1, b2, usB);
    };
    // the warped pixel pair of both images at row yw, given the up-sampled flow of the row (warp_kernel's map build)
    auto warp_row = [&](int yw, double uuA, double vvA, double uuB, double vvB, double &w1a, double &w1b, double &w2a,
                        double &w2b) __attribute__((always_inline)) {
        const float yf = (float)yw;
        const float hxA = (float)uuA / 2, hyA = (float)vvA / 2, hxB = (float)uuB / 2, hyB = (float)vvB / 2;
#if LK_W3F_UNCOND
        // both images of a column gathered together, unconditionally (lk_device.h remap_bilinear_x2: bit-identical): two memory
        // round trips per marched row instead of four
#if LK_W3F_UNCOND == 2
        {   // all eight gathers of the row in flight at once
            const RemapTaps t1a = remap_taps(ip, H, W, xfa + hxA, yf + hyA), t2a = remap_taps(ip, H, W, xfa - hxA, yf - hyA);
            const RemapTaps t1b = remap_taps(ip, H, W, xfb + hxB, yf + hyB), t2b = remap_taps(ip, H, W, xfb - hxB, yf - hyB);
            const lk_d2u g0 = *reinterpret_cast<const lk_d2u *>(I1 + t1a.off), g1 = *reinterpret_cast<const lk_d2u *>(I1 + t1a.off + ip);
            const lk_d2u g2 = *reinterpret_cast<const lk_d2u *>(I2 + t2a.off), g3 = *reinterpret_cast<const lk_d2u *>(I2 + t2a.off + ip);
            const lk_d2u g4 = *reinterpret_cast<const lk_d2u *>(I1 + t1b.off), g5 = *reinterpret_cast<const lk_d2u *>(I1 + t1b.off + ip);
            const lk_d2u g6 = *reinterpret_cast<const lk_d2u *>(I2 + t2b.off), g7 = *reinterpret_cast<const lk_d2u *>(I2 + t2b.off + ip);
            w1a = remap_blend(t1a, g0, g1, H, W); w2a = remap_blend(t2a, g2, g3, H, W);
            w1b = remap_blend(t1b, g4, g5, H, W); w2b = remap_blend(t2b, g6, g7, H, W);
        }
#else
        remap_bilinear_x2(I1, I2, ip, H, W, xfa + hxA, yf + hyA, xfa - hxA, yf - hyA, w1a, w2a);
        remap_bilinear_x2(I1, I2, ip, H, W, xfb + hxB, yf + hyB, xfb - hxB, yf - hyB, w1b, w2b);
#endif
#else
        w1a = remap_bilinear(I1, ip, H, W, xfa + hxA, yf + hyA);
        w2a = remap_bilinear(I2, ip, H, W, xfa - hxA, yf - hyA);
        w1b = remap_bilinear(I1, ip, H, W, xfb + hxB, yf + hyB);
        w2b = remap_bilinear(I2, ip, H, W, xfb - hxB, yf - hyB);
#endif
    };
    // cv::pyrUp at an arbitrary row (prologue only): vertical combination of three freshly fetched coarse rows
    auto up_at = [&](int yw, double &uuA, double &vvA, double &uuB, double &vvB) __attribute__((always_inline)) {
        const int k = yw >> 1;
        double u0A, u0B, v0A, v0B, u1A, u1B, v1A, v1B, u2A, u2B, v2A, v2B;
        hrow(k - 1, u0A, u0B, v0A, v0B); hrow(k, u1A, u1B, v1A, v1B); hrow(k + 1, u2A, u2B, v2A, v2B);
        if (yw & 1) {
            uuA = ((u1A + u2A) * 4) * (1.0 / 64); uuB = ((u1B + u2B) * 4) * (1.0 / 64);
            vvA = ((v1A + v2A) * 4) * (1.0 / 64); vvB = ((v1B + v2B) * 4) * (1.0 / 64);
        } else {
            uuA = (u0A + u1A * 6 + u2A) * (1.0 / 64); uuB = (u0B + u1B * 6 + u2B) * (1.0 / 64);
            vvA = (v0A + v1A * 6 + v2A) * (1.0 / 64); vvB = (v0B + v1B * 6 + v2B) * (1.0 / 64);
        }
    };
    auto left_tap = [&](double vb) __attribute__((always_inline)) {
        const double l = dpp_wave_shr1(vb);
        return at_x0 ? vb : l;
    };

    // ---- prologue: warped rows yy0-1 (previous) and yy0 (current), reflected into the image like cv::filter2D does ----
    double p1a, p1b, p2a, p2b, c1a, c1b, c2a, c2b;
    {
        double uuA, vvA, uuB, vvB;
        const int rp = r101(yy0 - 1, H), rc = r101(yy0, H);
        up_at(rp, uuA, vvA, uuB, vvB); warp_row(rp, uuA, vvA, uuB, vvB, p1a, p1b, p2a, p2b);
        up_at(rc, uuA, vvA, uuB, vvB); warp_row(rc, uuA, vvA, uuB, vvB, c1a, c1b, c2a, c2b);
    }
    double lp1 = left_tap(p1b), lp2 = left_tap(p2b), lc1 = left_tap(c1b), lc2 = left_tap(c2b);

    // carried coarse rows for row yw = y0 + i (the row produced in iteration i): s(iw-1), s(iw), s(iw+1), iw = yw >> 1
    double huA[3], huB[3], hvA[3], hvB[3];
    {
        const int i0 = y0 >> 1;
#pragma unroll
        for (int s = 0; s < 3; ++s) hrow(i0 - 1 + s, huA[s], huB[s], hvA[s], hvB[s]);
    }
    double tq3[3][3];                                            // coarse taps in flight (TAP3 scheme of lk_march3.h)
    auto tap_row = [&](int y) { return max(up_srow(((y + 1) >> 1) + 1) - crow_lo, 0); };
    {   // taps of rows y0-1 (slot 2) and y0 (slot 0), as iterations -2 and -1 (chunk positions 1 and 2) would have fetched them
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int yq = y0 - 1 + d;
            const bool odd = (yq & 1) != 0;
            const __amdgpu_buffer_rsrc_t rs = odd ? rCV : rCU;
            const int so = tap_row(yq | 1) * (odd ? cpvb : cpub);
            const int sl = d == 0 ? 2 : 0;
            tq3[sl][0] = buf_ld(rs, usA.jm, so); tq3[sl][1] = buf_ld(rs, usA.jc, so); tq3[sl][2] = buf_ld(rs, usA.jp, so);
        }
        tq3[1][0] = tq3[1][1] = tq3[1][2] = 0.0;
    }
    // up-sampled flow of the two previous produced rows (delay line): d1* = row yw-1, d2* = row yw-2
    double d1uA = 0, d1vA = 0, d1uB = 0, d1vB = 0, d2uA = 0, d2vA = 0, d2uB = 0, d2vB = 0;

    const int upb = (int)a.u.pitch * 8, vpb = (int)a.v.pitch * 8;
    const __amdgpu_buffer_rsrc_t rU = buf_rsrc(a.u.at(pair) + (long long)y0 * a.u.pitch, ((long long)(y1 - 1 - y0) * a.u.pitch + W) * 8);
    const __amdgpu_buffer_rsrc_t rV = buf_rsrc(a.v.at(pair) + (long long)y0 * a.v.pitch, ((long long)(y1 - 1 - y0) * a.v.pitch + W) * 8);
    const int vxo = xa * 8;

    auto row = [&](auto kc, int i) __attribute__((always_inline)) {
        constexpr int K = decltype(kc)::value;
        const int yy = yy0 + i;                                 // row whose products are formed now (current row c*)
        const int yw = yy + 1;                                  // row produced now (becomes c* at the rotate) = y0 + i
        const bool row_in = yy >= 0 && yy < H;
        const int y = yy - R;                                   // output row of this iteration = yw - 2
        const bool out_row = i >= 2 * R;
        {   // three coarse taps per row, one row ahead of the pyrUp that consumes them (yq = yw + 1)
            const int yq = yw + 1;
            const bool odd = (yq & 1) != 0;
            const __amdgpu_buffer_rsrc_t rs = odd ? rCV : rCU;
            const int so = tap_row(yq | 1) * (odd ? cpvb : cpub);
            constexpr int ts = (K + 1) % 3;
            tq3[ts][0] = buf_ld(rs, usA.jm, so); tq3[ts][1] = buf_ld(rs, usA.jc, so); tq3[ts][2] = buf_ld(rs, usA.jp, so);
        }
        // cv::pyrUp of the coarse flow at row yw (same schedule as the flow term of lk_w3x2_kernel, two rows earlier)
        double uuA, vvA, uuB, vvB;
        if ((yw & 1) == 0 && yw > y0) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                double *hu = c ? huB : huA, *hv = c ? hvB : hvA;
                const double nu = hu[0], nv = hv[0];
                hu[0] = hu[1]; hu[1] = hu[2]; hu[2] = nu;
                hv[0] = hv[1]; hv[1] = hv[2]; hv[2] = nv;
            }
        }
        if (yw & 1) {
            uuA = ((huA[1] + huA[2]) * 4) * (1.0 / 64); vvA = ((hvA[1] + hvA[2]) * 4) * (1.0 / 64);
            uuB = ((huB[1] + huB[2]) * 4) * (1.0 / 64); vvB = ((hvB[1] + hvB[2]) * 4) * (1.0 / 64);
            // next coarse row (u taps came with the previous, even, row -- slot K+2 --, v taps with this one -- slot K)
            huA[0] = pyrup_h1v(tq3[(K + 2) % 3][0], tq3[(K + 2) % 3][1], tq3[(K + 2) % 3][2], usA);
            huB[0] = pyrup_h1v(tq3[(K + 2) % 3][0], tq3[(K + 2) % 3][1], tq3[(K + 2) % 3][2], usB);
            hvA[0] = pyrup_h1v(tq3[K % 3][0], tq3[K % 3][1], tq3[K % 3][2], usA);
            hvB[0] = pyrup_h1v(tq3[K % 3][0], tq3[K % 3][1], tq3[K % 3][2], usB);
        } else {
            uuA = (huA[0] + huA[1] * 6 + huA[2]) * (1.0 / 64); vvA = (hvA[0] + hvA[1] * 6 + hvA[2]) * (1.0 / 64);
            uuB = (huB[0] + huB[1] * 6 + huB[2]) * (1.0 / 64); vvB = (hvB[0] + hvB[1] * 6 + hvB[2]) * (1.0 / 64);
        }
        // produce the warped row yw (rows below the image are never used: their products are zeroed)
        double n1a, n1b, n2a, n2b;
        warp_row(min(yw, H - 1), uuA, vvA, uuB, vvB, n1a, n1b, n2a, n2b);

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
            SA[f] = sum9_step(w9A[f], dpp_wave_shr1(pB[f]), pA[f], pB[f]);   // get_Sum9's own order (lk_device.h)
            SB[f] = sum9_step(w9B[f], pA[f], pB[f], dpp_wave_shl1(pA[f]));
        }
        p1a = c1a; p1b = c1b; p2a = c2a; p2b = c2b; lp1 = lc1; lp2 = lc2;
        c1a = n1a; c1b = n1b; c2a = n2a; c2b = n2b;
        lc1 = left_tap(c1b); lc2 = left_tap(c2b);

        if (out_row && is_out) {                                // flow of row y = yw - 2: its pyrUp term sits in d2*
            double uA = 0.0, vA = 0.0, uB = 0.0, vB = 0.0;
            const bool inner_row = y >= 1 && y <= H - 2;
            if (inner_row && xa >= 1) lk_solve(SA, uA, vA);
            if (inner_row && xa + 1 <= W - 2) lk_solve(SB, uB, vB);
            buf_st2(rU, vxo, (y - y0) * upb, uA + d2uA, uB + d2uB);
            buf_st2(rV, vxo, (y - y0) * vpb, vA + d2vA, vB + d2vB);
        }
        d2uA = d1uA; d2vA = d1vA; d2uB = d1uB; d2vB = d1vB;
        d1uA = uuA; d1vA = vvA; d1uB = uuB; d1vB = vvB;
    };

    for (int base = 0; base < nmarch; base += WIN) {
        [&]<int... Ks>(std::integer_sequence<int, Ks...>) __attribute__((always_inline)) {
            ((base + Ks < nmarch ? (row(std::integral_constant<int, Ks>{}, base + Ks), 0) : 0), ...);
        }(std::make_integer_sequence<int, WIN>{});
    }
}

}  // namespace lk
