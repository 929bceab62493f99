// lk_batch.h -- single-level dense LK (a3-a8) with the level's warp (a9-a12) FUSED IN, row-marching, BATCH-synchronous:
// a workgroup of 8 identical waves marches a block of 126 columns down a row strip EIGHT rows at a time, and every
// wave does every kind of work (ParallelVideoPyr.cpp:646-703 warp, :44-84 derivatives, :114-191 window sums,
// :470-520 solve, :634-637 accumulate).
//
// Why (round 3): the wave-specialised kernel (lk_split.h) runs its three roles in lock-step, one workgroup barrier
// per two rows, with one "F" wave feeding seven; two thirds of its time is that cadence, and a fourth role (the warp)
// does not fit a CU twice.  Here a batch of 8 marched rows goes through two phases per workgroup:
//
//   phase A, all 8 waves, wave w <-> row w of the batch, one column per lane, two half-rows one after the other:
//     W part  the warped pixel pair of the row (FW: cv::pyrUp of the coarse flow at the pixel, float map, 1/32-pixel
//             fixed-point bilinear remap of both UN-warped level images -- warp_kernel's arithmetic; !FW: a plain
//             load) -> s = I1w + I2w and d = I2w - I1w -> LDS rows (the warped planes never reach memory);
//     B part  the five window sums of one row of the PREVIOUS batch come back from LDS -> 2 x 2 solve with true fp64
//             divisions, /0 -> 0, "+= pyrUp(coarse flow)", streaming store of u, v;
//   phase S, waves 0..4, one FIELD each (fx^2, fy^2, fx fy, fx ft, fy ft), two adjacent columns per lane, the 8 rows
//     of the batch one after the other: the two derivatives of the field from the (s, d) rows (left tap by a DPP wave
//     shift, previous row's half in registers) -> product -> horizontal window: the lane's pair, its sum and the
//     neighbours' through a wave-PRIVATE LDS row (no barrier: a wave's LDS instructions execute in order) ->
//     vertical window: van Herk / Gil-Werman prefix + suffix ring in registers -> the row's two window sums -> LDS.
//
// Two workgroup barriers per EIGHT rows (lk_split.h: four), every wave with the same instruction stream in phase A,
// five of eight busy in phase S; two workgroups per CU (<= 128 VGPRs, 66 KB LDS) run in different phases.
//
// Arithmetic.  Window sums use additions only (flat regions give exactly 0 like the reference).  The derivatives are
// evaluated separably on s = I1 + I2 and d = I2 - I1:
//     fx = (s(y,x) - s(y,x-1)) + (s(y-1,x) - s(y-1,x-1))        [filter2D Kx on I1 + filter2D Kx on I2, :44-56]
//     fy = (s(y,x-1) + s(y,x)) - (s(y-1,x-1) + s(y-1,x))        [:58-70]
//     ft = (d(y,x-1) + d(y,x)) + (d(y-1,x-1) + d(y-1,x))        [:72-84]
// i.e. the reference's eight-term sums re-associated (each S wave needs its derivatives itself, and the separable form
// costs 6 instead of 14 additions per derivative): results differ from lk_split.h / the oracle in the last bits of the
// derivatives (a few 1e-16 of the pixel range), far inside the reference's own 1e-6 bar (ParallelBasicPyr.cpp:787);
// identical frames still give ft == 0 exactly and flat regions fx == fy == 0 exactly.
// The vertical ring uses chunks of L = WIN + 1 rows anchored to ABSOLUTE image rows (window = suffix of the previous
// chunk + prefix of the current one; the two windows that lie inside one chunk are prefix[WIN-1] and suffix[1]), so
// that a result does not depend on the strip height the launcher picks (sharded == unsharded bit for bit) and the
// chunk position of a batch row is static: the body is unrolled over lcm(L, 8) / 8 batch variants.
//
// The frame of width R just inside the image (clipped windows m(y,x) * WIN^2, SerialPyr.cpp:94-95, and the outermost
// ring) is written by lk_border_pixel (lk_split.h) in trailing workgroups of the same launch.
// Host-checked: W and H even, strip height a multiple of 8.
#pragma once
#include "lk_split.h"

// compiler-only ordering point for the wave-private LDS rows (a wave's LDS instructions execute in issue order)
#define BT_WAVE_FENCE() asm volatile("" ::: "memory")

namespace lk {

constexpr int bt_gcd(int a, int b) { return b == 0 ? a : bt_gcd(b, a % b); }

template <int WIN>
struct BatchCfg {
    static constexpr int R = WIN / 2;
    static constexpr int L = WIN + 1;                  // van Herk chunk length (rows)
    static constexpr int G = 8;                        // rows per batch = waves per workgroup
    static constexpr int NV = L / bt_gcd(L, G);        // batch variants in the unrolled S body: lcm(L, 8) / 8
    static constexpr int HL = R + (R & 1);             // halo columns on each side (even)
    static constexpr int XO = 126 - 2 * HL;            // output columns per workgroup
    static constexpr int NT = 64 * G;
    static constexpr int APR = (R + 1 + 7) & ~7;       // marched rows above the strip's first output row (multiple of 8)
    static constexpr bool RODD = (R & 1) != 0;
    static constexpr int HS = RODD ? (R + 1) / 2 : R / 2;   // lane offset of the two single products of a window pair
    static constexpr int QH = RODD ? (R - 1) / 2 : R / 2;   // pair sums are read for lane offsets -QH .. +QH
    static constexpr int NQ = 2 * QH + 1;
    static constexpr int PAD = HS;                     // pad entries on each side of a private row
    static constexpr int NCH = 64 + 2 * PAD;
    static_assert(WIN >= 5 && WIN <= 15 && (WIN & 1), "ring of WIN x 2 x 2 VGPRs next to the phase-A code: <= 128 VGPRs up to 15 x 15");
};

template <int WIN, bool HAS_UP, bool FW>
__global__ __launch_bounds__(512, 4) void lk_batch_kernel(LKArgs a)
{
    using C = BatchCfg<WIN>;
    static_assert(!FW || HAS_UP, "the fused warp needs a coarser level");
    constexpr int R = C::R, L = C::L, HL = C::HL, XO = C::XO, HS = C::HS, QH = C::QH, NQ = C::NQ, PAD = C::PAD;
    if (a.nmain && blockIdx.x >= a.nmain) {
        // trailing workgroups of the launch: the frame of width R just inside the image, one pixel per thread
        const unsigned bb = blockIdx.x - a.nmain;
        lk_border_pixel<HAS_UP, FW>(a, WIN, (bb % a.nbb) * (unsigned)C::NT + threadIdx.x, (int)(bb / a.nbb));
        return;
    }
    __shared__ __attribute__((aligned(16))) double s_im[8][2][128];      // [batch row][image][block column + 2]: the (warped) rows
    __shared__ __attribute__((aligned(16))) double s_prev[2][2][128];    // [batch parity][image]: row 7 of batch n, the row above batch n + 1
    __shared__ __attribute__((aligned(16))) double s_sums[8][5][128];    // [batch row][field][block column + 2]
    __shared__ double s_pa[5][C::NCH], s_pb[5][C::NCH], s_pq[5][C::NCH]; // wave-private rows: products a, b of every pair, a + b

    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lane = (int)threadIdx.x & 63;
    unsigned tile = xcd_tile_id(blockIdx.x, a.nmain ? a.nmain : gridDim.x);
    const int bx = tile % a.gx; tile /= a.gx;
    const int by = tile % a.gy;
    const int pair = tile / a.gy;
    const int H = a.H, W = a.W;
    const int y0 = by * a.th;                          // multiple of 8 (host)
    const int y1 = min(y0 + a.th, H);
    const int ym0 = y0 - C::APR;                       // image row of marched row 0 (multiple of 8)
    const int x0 = bx * XO - HL;                       // image column of block column 0 (even); LDS column c = block column + 2
    const int nmr = y1 - 1 + R - ym0 + 1;              // marched rows: image rows ym0 .. y1-1+R
    const int NBS = (nmr + 7) >> 3;                    // S batches
    const int ylast = min(y1 - 1 + R, H - 1);          // last image row whose products an output needs
    const int yo_lo = max(y0, R), yo_hi = min(y1, H - R);   // output rows this workgroup stores (the frame is lk_border_pixel's)

    const double *I1 = a.i1.at(pair), *I2 = a.i2.at(pair);
    const int ip = (int)a.i1.pitch;                    // (host: one plane < 2^31 elements, both images one pitch)
    [[maybe_unused]] const double *cu = HAS_UP ? a.cu.at(pair) : nullptr, *cv = HAS_UP ? a.cv.at(pair) : nullptr;
    [[maybe_unused]] const int Hc = H >> 1, Wc = W >> 1;
    double *Uo = a.u.at(pair), *Vo = a.v.at(pair);

    // phase A geometry: wave wv <-> rows (2k, 2k+1) of the batch, k = wv >> 1, columns 64 (wv & 1) + lane
    const int ak = wv >> 1;
    const int ac = 64 * (wv & 1) + lane;               // LDS column
    const int ax = x0 - 2 + ac;                        // image column
    const bool a_out = ac >= HL + 2 && ac < HL + 2 + XO && ax >= R && ax < W - R;   // an output column of this workgroup, not in the frame
    // column -1 is filter2D's reflect-101 tap of column 0 = column 1; other columns outside the image only feed products
    // that S zeroes (any in-image column will do)
    const int axw = min(ax < 0 ? -ax : ax, W - 1);

    // cv::pyrUp of the coarse flow at the row pair (2i, 2i+1), column x, straight-line (make_upsel1 / pyrup_h1v: pyrup_at()'s
    // roundings): [0] the even row, [1] the odd row
    [[maybe_unused]] auto upflow2 = [&](int i, int x, double (&uu)[2], double (&vv)[2]) __attribute__((always_inline)) {
        const UpSel1 us = make_upsel1(x, Wc);
        double hu[3], hv[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int k = i - 1 + q;
            const int sr = k < 0 ? min(1, Hc - 1) : (k > Hc - 1 ? Hc - 1 : k);
            const char *pu = reinterpret_cast<const char *>(cu + (long long)sr * a.cu.pitch);
            const char *pv = reinterpret_cast<const char *>(cv + (long long)sr * a.cv.pitch);
            hu[q] = pyrup_h1v(*reinterpret_cast<const double *>(pu + us.jm), *reinterpret_cast<const double *>(pu + us.jc),
                              *reinterpret_cast<const double *>(pu + us.jp), us);
            hv[q] = pyrup_h1v(*reinterpret_cast<const double *>(pv + us.jm), *reinterpret_cast<const double *>(pv + us.jc),
                              *reinterpret_cast<const double *>(pv + us.jp), us);
        }
        uu[0] = (hu[0] + hu[1] * 6 + hu[2]) * (1.0 / 64); vv[0] = (hv[0] + hv[1] * 6 + hv[2]) * (1.0 / 64);
        uu[1] = ((hu[1] + hu[2]) * 4) * (1.0 / 64);       vv[1] = ((hv[1] + hv[2]) * 4) * (1.0 / 64);
    };

    // ---- phase A, W part: the wave's row pair of batch n -> LDS ------------------------------------------------------
    auto wpart = [&](int n) __attribute__((always_inline)) {
        const int ye = ym0 + 8 * n + 2 * ak;           // even image row of the pair (ym0 is even)
        if (ye + 1 < -1 || ye > ylast) return;         // (wave-uniform) rows no output needs: S zeroes their products
        // row -1 is filter2D's reflect-101 tap of row 0 = row 1 (rows -2 and below, H and beyond: any in-image row will do)
        const int yw0 = min(ye < 0 ? 1 : ye, H - 1), yw1 = min(ye + 1 < 0 ? 1 : ye + 1, H - 1);
        double w1[2], w2[2];
        if constexpr (FW) {
            double uu[2], vv[2];
            upflow2(yw0 >> 1, axw, uu, vv);
            if (ye < 0) { uu[0] = uu[1]; vv[0] = vv[1]; }   // (wave-uniform) the pair (-2, -1) -> rows (1, 1): the odd row of coarse pair 0
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int yw = e ? yw1 : yw0;
                const float hx = (float)uu[e] / 2, hy = (float)vv[e] / 2;   // map build (ParallelVideoPyr.cpp:674-678), float32
                w1[e] = remap_bilinear(I1, ip, H, W, (float)axw + hx, (float)yw + hy);
                w2[e] = remap_bilinear(I2, ip, H, W, (float)axw - hx, (float)yw - hy);
                // the up-sampled flow of an output pixel is parked in U, V: the B part adds it to the solve two batches on
                if (a_out && ye + e >= yo_lo && ye + e < yo_hi) {
                    Uo[(long long)(ye + e) * a.u.pitch + ax] = uu[e];
                    Vo[(long long)(ye + e) * a.v.pitch + ax] = vv[e];
                }
            }
        } else {
            w1[0] = I1[(long long)yw0 * ip + axw]; w2[0] = I2[(long long)yw0 * ip + axw];
            w1[1] = I1[(long long)yw1 * ip + axw]; w2[1] = I2[(long long)yw1 * ip + axw];
        }
        s_im[2 * ak][0][ac] = w1[0]; s_im[2 * ak][1][ac] = w2[0];
        s_im[2 * ak + 1][0][ac] = w1[1]; s_im[2 * ak + 1][1][ac] = w2[1];
        if (ak == 3) { s_prev[n & 1][0][ac] = w1[1]; s_prev[n & 1][1][ac] = w2[1]; }   // (wave-uniform)
    };

    // ---- phase A, B part: the wave solves its row pair of the sums S(n - 1) left behind -----------------------------
    auto bpart = [&](int n) __attribute__((always_inline)) {
        const int yb = ym0 + 8 * (n - 1) + 2 * ak - R; // image row of sums row 2 ak (the pair: yb, yb + 1)
        if (yb + 1 < yo_lo || yb >= yo_hi) return;      // (wave-uniform)
        [[maybe_unused]] double uu[2] = {0.0, 0.0}, vv[2] = {0.0, 0.0};
        if constexpr (HAS_UP && !FW) {
            // (R odd: the pair (yb, yb+1) is (odd, even) -- two coarse pairs; the unfused path is not the hot one)
            const int xq = min(max(ax, 0), W - 1);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int y = min(max(yb + e, 0), H - 1);
                double tu[2], tv[2];
                upflow2(y >> 1, xq, tu, tv);
                uu[e] = tu[y & 1]; vv[e] = tv[y & 1];
            }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int yo = yb + e;
            const bool st = a_out && yo >= yo_lo && yo < yo_hi;
            double S[5];
#pragma unroll
            for (int f = 0; f < 5; ++f) S[f] = s_sums[2 * ak + e][f][ac];
            double u, v;
            lk_solve(S, u, v);
            if (st) {
                double *pu = Uo + (long long)yo * a.u.pitch + ax, *pv = Vo + (long long)yo * a.v.pitch + ax;
                if constexpr (FW) { u += *pu; v += *pv; }           // parked by the W part of this workgroup
                else if constexpr (HAS_UP) { u += uu[e]; v += vv[e]; }
                __builtin_nontemporal_store(u, pu);
                __builtin_nontemporal_store(v, pv);
            }
        }
    };

    // ---- phase S state (waves 0..4; the others carry it along unused) ---------------------------------------------
    const int fw = min(wv, 4);                         // field of this S wave: fx^2, fy^2, fx fy, fx ft, fy ft
    // derivative types of the field's two factors: 0 = fx, 1 = fy, 2 = ft.  One image's 2 x 2 filter response in the
    // reference's order (lk_device.h f2d_*): D = ((-p00 + c1 p01) + c2 p10) + c3 p11, the derivative = D(image 1) + sg D(image 2):
    //   fx: c = (+, -, +), sg = +;   fy: c = (-, +, +), sg = +;   ft: c = (-, -, -), sg = -  (f2d_pos(b) == -f2d_neg(b) exactly)
    // multiplying by +-1.0 inside an FMA rounds exactly where the reference's additions and subtractions round
    const int T1 = (fw == 1 || fw == 4) ? 1 : 0;                       // xx, xy, xt: fx;  yy, yt: fy
    const int T2 = fw == 0 ? 0 : (fw <= 2 ? 1 : 2);                    // xx: fx;  yy, xy: fy;  xt, yt: ft
    const double c11 = T1 == 0 ? 1.0 : -1.0, c12 = T1 == 1 ? 1.0 : -1.0, c13 = T1 == 2 ? -1.0 : 1.0, sg1 = T1 == 2 ? -1.0 : 1.0;
    const double c21 = T2 == 0 ? 1.0 : -1.0, c22 = T2 == 1 ? 1.0 : -1.0, c23 = T2 == 2 ? -1.0 : 1.0, sg2 = T2 == 2 ? -1.0 : 1.0;
    const double *im_row = &s_im[0][0][2 * lane];
    double *pa_row = &s_pa[fw][PAD + lane], *pb_row = &s_pb[fw][PAD + lane], *pq_row = &s_pq[fw][PAD + lane];
    double *sums_row = &s_sums[0][fw][2 * lane];
    const int xs = x0 - 2 + 2 * lane;                  // image column of the lane's first column (even: the pair is all in or all out)
    const bool col_in = xs >= 0 && xs < W;
    const bool all_col_in = __builtin_amdgcn_ballot_w64(!col_in) == 0;
    double ring[L][2], pre[2] = {0.0, 0.0};
#pragma unroll
    for (int q = 0; q < L; ++q) ring[q][0] = ring[q][1] = 0.0;
    // the previous marched row: left tap, column a, column b of image 1 (A) and image 2 (B); live inside a batch only
    double pAl, pAa, pAb, pBl, pBa, pBb;

    // one marched row at static chunk position KC, batch row RR
    auto srow = [&](auto kcc, auto rrc, int n) __attribute__((always_inline)) {
        constexpr int KC = decltype(kcc)::value, RR = decltype(rrc)::value;
        const int i = 8 * n + RR, yy = ym0 + i;
        const bool row_ok = yy >= 0 && yy < H && i >= 1;       // (wave-uniform) row 0 of the march has no previous row
        const lk_d2 mA = *reinterpret_cast<const lk_d2 *>(im_row + RR * 256);
        const lk_d2 mB = *reinterpret_cast<const lk_d2 *>(im_row + RR * 256 + 128);
        const double cAl = dpp_wave_shr1(mA.y), cBl = dpp_wave_shr1(mB.y);   // left tap of column a: the left lane's column b
        auto resp = [&](double k1, double k2, double k3, double p00, double p01, double p10, double p11) __attribute__((always_inline)) {
            return __builtin_fma(k3, p11, __builtin_fma(k2, p10, __builtin_fma(k1, p01, -p00)));
        };
        const double g1a = __builtin_fma(sg1, resp(c11, c12, c13, pBl, pBa, cBl, mB.x), resp(c11, c12, c13, pAl, pAa, cAl, mA.x));
        const double g1b = __builtin_fma(sg1, resp(c11, c12, c13, pBa, pBb, mB.x, mB.y), resp(c11, c12, c13, pAa, pAb, mA.x, mA.y));
        const double g2a = __builtin_fma(sg2, resp(c21, c22, c23, pBl, pBa, cBl, mB.x), resp(c21, c22, c23, pAl, pAa, cAl, mA.x));
        const double g2b = __builtin_fma(sg2, resp(c21, c22, c23, pBa, pBb, mB.x, mB.y), resp(c21, c22, c23, pAa, pAb, mA.x, mA.y));
        pAl = cAl; pAa = mA.x; pAb = mA.y; pBl = cBl; pBa = mB.x; pBb = mB.y;
        double pa = g1a * g2a, pb = g1b * g2b;
        if (!(row_ok && all_col_in)) {                  // rows / columns outside the image contribute 0 (wave-uniform test)
            const bool keep = row_ok && col_in;
            pa = keep ? pa : 0.0; pb = keep ? pb : 0.0;
        }
        // horizontal windows of the pair through the wave's private row (in-order LDS: no barrier, no wait)
        // (the fences are compiler-only: seen from ONE thread the stores and the neighbours' loads do not alias, so
        // nothing else keeps the compiler from moving a load above the store of the lane it reads from)
        pa_row[0] = pa; pb_row[0] = pb; pq_row[0] = pa + pb;
        BT_WAVE_FENCE();
        double q[NQ];
#pragma unroll
        for (int d = 0; d < NQ; ++d) q[d] = pq_row[d - QH];
        const double sb = pb_row[-HS], sa = pa_row[HS];
        BT_WAVE_FENCE();
        double hs[2];
        if constexpr (C::RODD) {
            const double shared = split_tree<0, NQ - 1>(q);
            hs[0] = sb + shared; hs[1] = shared + sa;
        } else {
            const double shared = split_tree<1, NQ - 2>(q);
            hs[0] = (q[0] + shared) + sa; hs[1] = (sb + shared) + q[NQ - 1];
        }
        // vertical window: van Herk ring, chunks of L = WIN + 1 rows
        lk_d2 o;
        double S[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            pre[c] = (KC == 0) ? hs[c] : pre[c] + hs[c];
            if constexpr (KC == L - 2) S[c] = pre[c];                       // rows 0 .. WIN-1 of this chunk
            else if constexpr (KC < L - 2) S[c] = ring[KC + 2][c] + pre[c]; // suffix of the previous chunk + prefix of this one
            if constexpr (KC >= 1) ring[KC][c] = hs[c];
            if constexpr (KC == L - 1) {                                    // chunk complete: suffix sums in place
#pragma unroll
                for (int qq = L - 2; qq >= 1; --qq) ring[qq][c] += ring[qq + 1][c];
                S[c] = ring[1][c];                                          // rows 1 .. WIN of this chunk
            }
        }
        o.x = S[0]; o.y = S[1];
        *reinterpret_cast<lk_d2 *>(sums_row + RR * 640) = o;
    };
    auto sbatch = [&](auto vc, int n) __attribute__((always_inline)) {
        constexpr int V = decltype(vc)::value;
        {   // the row above the batch (batch 0: stale memory -- row 0 of the march is zeroed anyway)
            const lk_d2 mA = *reinterpret_cast<const lk_d2 *>(&s_prev[(n + 1) & 1][0][2 * lane]);
            const lk_d2 mB = *reinterpret_cast<const lk_d2 *>(&s_prev[(n + 1) & 1][1][2 * lane]);
            pAl = dpp_wave_shr1(mA.y); pAa = mA.x; pAb = mA.y; pBl = dpp_wave_shr1(mB.y); pBa = mB.x; pBb = mB.y;
        }
        [&]<int... Rs>(std::integer_sequence<int, Rs...>) __attribute__((always_inline)) {
            (srow(std::integral_constant<int, (8 * V + Rs) % L>{}, std::integral_constant<int, Rs>{}, n), ...);
        }(std::make_integer_sequence<int, 8>{});
    };

    // zero the pads of the private rows once (only non-output lanes ever read them)
    for (int k = threadIdx.x; k < 5 * 2 * PAD; k += C::NT) {
        const int f = k / (2 * PAD), p = k % (2 * PAD);
        const int e = p < PAD ? p : 64 + p;
        s_pa[f][e] = 0.0; s_pb[f][e] = 0.0; s_pq[f][e] = 0.0;
    }

    constexpr int LCM = 8 * C::NV;
    int v = (((ym0 % LCM) + LCM) % LCM) >> 3;          // variant of batch 0: chunk position of marched row 0 = 8 v mod L
    for (int n = 0; n <= NBS; ++n) {
        if (n < NBS) wpart(n);
        if (n >= 1) bpart(n);
        __syncthreads();                               // rows of batch n complete; sums of batch n - 1 consumed
        if (n < NBS && wv < 5) {
            [&]<int... Vs>(std::integer_sequence<int, Vs...>) __attribute__((always_inline)) {
                ((v == Vs ? (sbatch(std::integral_constant<int, Vs>{}, n), 0) : 0), ...);
            }(std::make_integer_sequence<int, C::NV>{});
        }
        v = v + 1 == C::NV ? 0 : v + 1;
        __syncthreads();                               // sums of batch n complete; rows free
    }
}

}  // namespace lk
