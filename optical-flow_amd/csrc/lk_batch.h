// lk_batch.h -- single-level dense LK (a3-a8) with the level's warp (a9-a12) FUSED IN, row-marching, BATCH-synchronous:
// a workgroup of 8 waves marches a block of 126 columns down a row strip EIGHT rows at a time
// (ParallelVideoPyr.cpp:646-703 warp, :44-84 derivatives, :114-191 window sums, :470-520 solve, :634-637 accumulate).
//
// Why (round 3): the wave-specialised kernel (lk_split.h) runs its three roles in lock-step, one workgroup barrier
// per two rows, with one "F" wave feeding seven; two thirds of its time is that cadence, and a fourth role (the warp)
// does not fit a CU twice.  Here the synchronisation is per batch of 8 marched rows, two barriers per batch:
//
//   phase X(n)   waves 0..4 ("S", one FIELD each: fx^2, fy^2, fx fy, fx ft, fy ft; two adjacent columns per lane) take
//                the 8 rows of batch n, two rows per LDS round trip: the two derivatives of the field in the
//                reference's own expression order from the (warped) image rows in LDS (left tap by a DPP wave shift,
//                previous row in registers) -> product -> horizontal window: the lane's pair, its sum and the
//                neighbours' through a wave-PRIVATE LDS row (no barrier: a wave's LDS instructions execute in order)
//                -> vertical window: van Herk / Gil-Werman prefix + suffix ring in registers -> window sums -> LDS;
//                MEANWHILE waves 5..7 ("W", one column per lane, units of a row pair x 64 columns) make the rows of
//                batch n + 1 into the other half of a double-buffered LDS row store: FW: cv::pyrUp of the coarse flow
//                at the pixel (parked in U, V for phase B), float map, 1/32-pixel fixed-point bilinear remap of both
//                UN-warped level images (warp_kernel's arithmetic; the warped planes never reach memory); !FW: plain
//                loads (+ the parked pyrUp if there is a coarser level).  All of W's memory latency hides behind S.
//   phase B(n)   all 8 waves, one unit each: the five window sums of a row pair come back from LDS -> 2 x 2 solve with
//                true fp64 divisions, /0 -> 0, "+= pyrUp(coarse flow)" (parked by W), streaming store of u, v.
//
// Two workgroup barriers per EIGHT rows (lk_split.h: four); two workgroups per CU (<= 128 VGPRs, 76 KB LDS).
//
// Arithmetic: lk_device.h's expressions -- the 2 x 2 filter2D responses in OpenCV's tap order (an S wave evaluates the
// one or two derivatives of ITS field: ((-p00 + c1 p01) + c2 p10) + c3 p11 with c = +-1.0 inside FMAs, which rounds
// exactly where the reference's additions and subtractions round; f2d_pos(b) = -f2d_neg(b) exactly) -- then window
// sums with additions only (flat regions give exactly 0 like the reference).  This matters: where a window holds nothing
// but the rounding residue of the reference's own derivative order, that residue IS the reference's flow
// (tests/test_gpu_parity.py::test_random_small_shapes_against_oracle, flat half-images).
// The vertical ring uses chunks of L = WIN + 1 rows anchored to ABSOLUTE image rows (window = suffix of the previous
// chunk + prefix of the current one; the two windows that lie inside one chunk are prefix[WIN-1] and suffix[1]), so
// that a result does not depend on the strip height the launcher picks (sharded == unsharded bit for bit) and the
// chunk position of a batch row is static: the body is unrolled over lcm(L, 8) / 8 batch variants.
//
// The frame of width R just inside the image (clipped windows m(y,x) * WIN^2, SerialPyr.cpp:94-95, and the outermost
// ring) is written by lk_border_pixel (lk_split.h) in trailing workgroups of the same launch.
// Host-checked: W and H even, strip height a multiple of 8, both images one pitch.
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
    __shared__ __attribute__((aligned(16))) double s_im[2][8][2][128];   // [batch parity][batch row][image][block column + 2]
    __shared__ __attribute__((aligned(16))) double s_sums[8][5][XO];     // [batch row][field][output column of the workgroup]
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
    const int NBS = (nmr + 7) >> 3;                    // batches
    const int ylast = min(y1 - 1 + R, H - 1);          // last image row whose products an output needs
    const int yo_lo = max(y0, R), yo_hi = min(y1, H - R);   // output rows this workgroup stores (the frame is lk_border_pixel's)

    const double *I1 = a.i1.at(pair), *I2 = a.i2.at(pair);
    const int ip = (int)a.i1.pitch;                    // (host: one plane < 2^31 elements, both images one pitch)
    [[maybe_unused]] const double *cu = HAS_UP ? a.cu.at(pair) : nullptr, *cv = HAS_UP ? a.cv.at(pair) : nullptr;
    [[maybe_unused]] const int Hc = H >> 1, Wc = W >> 1;
    double *Uo = a.u.at(pair), *Vo = a.v.at(pair);

    // a UNIT = the row pair (2k, 2k+1) of a batch x 64 columns, one column per lane: unit u = 2 k + half
    struct Unit { int k, c, x, xw; bool out; };
    auto unit = [&](int u) __attribute__((always_inline)) {
        Unit t;
        t.k = u >> 1;
        t.c = 64 * (u & 1) + lane;                     // LDS column
        t.x = x0 - 2 + t.c;                            // image column
        t.out = t.c >= HL + 2 && t.c < HL + 2 + XO && t.x >= R && t.x < W - R;   // an output column of this workgroup, not in the frame
        // column -1 is filter2D's reflect-101 tap of column 0 = column 1; other columns outside the image only feed
        // products that S zeroes (any in-image column will do)
        t.xw = min(t.x < 0 ? -t.x : t.x, W - 1);
        return t;
    };

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

    // ---- W: one unit of batch n -> LDS rows -------------------------------------------------------------------------
    auto wunit = [&](int n, int u) __attribute__((always_inline)) {
        const Unit t = unit(u);
        const int ye = ym0 + 8 * n + 2 * t.k;          // even image row of the pair (ym0 is even)
        if (ye + 1 < -1 || ye > ylast) return;         // (wave-uniform) rows no output needs: S zeroes their products
        // row -1 is filter2D's reflect-101 tap of row 0 = row 1 (rows -2 and below, H and beyond: any in-image row will do)
        const int yw0 = min(ye < 0 ? 1 : ye, H - 1), yw1 = min(ye + 1 < 0 ? 1 : ye + 1, H - 1);
        double w1[2], w2[2];
        [[maybe_unused]] double uu[2], vv[2];
        if constexpr (HAS_UP) {
            upflow2(yw0 >> 1, t.xw, uu, vv);
            if (ye < 0) { uu[0] = uu[1]; vv[0] = vv[1]; }   // (wave-uniform) the pair (-2, -1) -> rows (1, 1): the odd row of coarse pair 0
            // the up-sampled flow of an output pixel is parked in U, V: phase B adds it to the solve two batches on
#pragma unroll
            for (int e = 0; e < 2; ++e)
                if (t.out && ye + e >= yo_lo && ye + e < yo_hi) {
                    Uo[(long long)(ye + e) * a.u.pitch + t.x] = uu[e];
                    Vo[(long long)(ye + e) * a.v.pitch + t.x] = vv[e];
                }
        }
        if constexpr (FW) {
            // map build (ParallelVideoPyr.cpp:674-678) in float32; all eight gathers of the unit unconditionally and up front
            // (lk_device.h remap_taps / remap_blend: remap_bilinear's arithmetic without its conditional loads)
            const float hx0 = (float)uu[0] / 2, hy0 = (float)vv[0] / 2, hx1 = (float)uu[1] / 2, hy1 = (float)vv[1] / 2;
            const float xf = (float)t.xw, yf0 = (float)yw0, yf1 = (float)yw1;
            const RemapTaps ta0 = remap_taps(ip, H, W, xf + hx0, yf0 + hy0), tb0 = remap_taps(ip, H, W, xf - hx0, yf0 - hy0);
            const RemapTaps ta1 = remap_taps(ip, H, W, xf + hx1, yf1 + hy1), tb1 = remap_taps(ip, H, W, xf - hx1, yf1 - hy1);
            const lk_d2u g0 = *reinterpret_cast<const lk_d2u *>(I1 + ta0.off), g1 = *reinterpret_cast<const lk_d2u *>(I1 + ta0.off + ip);
            const lk_d2u g2 = *reinterpret_cast<const lk_d2u *>(I2 + tb0.off), g3 = *reinterpret_cast<const lk_d2u *>(I2 + tb0.off + ip);
            const lk_d2u g4 = *reinterpret_cast<const lk_d2u *>(I1 + ta1.off), g5 = *reinterpret_cast<const lk_d2u *>(I1 + ta1.off + ip);
            const lk_d2u g6 = *reinterpret_cast<const lk_d2u *>(I2 + tb1.off), g7 = *reinterpret_cast<const lk_d2u *>(I2 + tb1.off + ip);
            w1[0] = remap_blend(ta0, g0, g1, H, W); w2[0] = remap_blend(tb0, g2, g3, H, W);
            w1[1] = remap_blend(ta1, g4, g5, H, W); w2[1] = remap_blend(tb1, g6, g7, H, W);
        } else {
            w1[0] = I1[(long long)yw0 * ip + t.xw]; w2[0] = I2[(long long)yw0 * ip + t.xw];
            w1[1] = I1[(long long)yw1 * ip + t.xw]; w2[1] = I2[(long long)yw1 * ip + t.xw];
        }
        s_im[n & 1][2 * t.k][0][t.c] = w1[0]; s_im[n & 1][2 * t.k][1][t.c] = w2[0];
        s_im[n & 1][2 * t.k + 1][0][t.c] = w1[1]; s_im[n & 1][2 * t.k + 1][1][t.c] = w2[1];
    };
    auto wbatch = [&](int n) __attribute__((always_inline)) {     // W wave wv - 5 takes units wv - 5, wv - 2, wv + 1 (< 8)
        for (int u = wv - 5; u < 8; u += 3) wunit(n, u);
    };

    // ---- phase B: the wave solves unit wv of the sums S(n) left behind ----------------------------------------------
    auto bphase = [&](int n) __attribute__((always_inline)) {
        const Unit t = unit(wv);
        const int yb = ym0 + 8 * n + 2 * t.k - R;      // image row of sums row 2 k (the pair: yb, yb + 1)
        if (yb + 1 < yo_lo || yb >= yo_hi) return;      // (wave-uniform)
        const int sc = min(max(t.c - (HL + 2), 0), XO - 1);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int yo = yb + e;
            const bool st = t.out && yo >= yo_lo && yo < yo_hi;
            double S[5];
#pragma unroll
            for (int f = 0; f < 5; ++f) S[f] = s_sums[2 * t.k + e][f][sc];
            double u, v;
            lk_solve(S, u, v);
            if (st) {
                double *pu = Uo + (long long)yo * a.u.pitch + t.x, *pv = Vo + (long long)yo * a.v.pitch + t.x;
                if constexpr (HAS_UP) { u += *pu; v += *pv; }       // parked by the W waves of this workgroup
                __builtin_nontemporal_store(u, pu);
                __builtin_nontemporal_store(v, pv);
            }
        }
    };

    // zero the pads of the private rows once (only non-output lanes ever read them)
    for (int k = threadIdx.x; k < 5 * 2 * PAD; k += C::NT) {
        const int f = k / (2 * PAD), p = k % (2 * PAD);
        const int e = p < PAD ? p : 64 + p;
        s_pa[f][e] = 0.0; s_pb[f][e] = 0.0; s_pq[f][e] = 0.0;
    }

    if (wv >= 5) {
        // =============================================================================================================
        // W waves
        // =============================================================================================================
        wbatch(0);
        __syncthreads();
        for (int n = 0; n < NBS; ++n) {
            if (n + 1 < NBS) wbatch(n + 1);
            __syncthreads();                           // sums of batch n and rows of batch n + 1 complete
            bphase(n);
            __syncthreads();                           // sums consumed
        }
        return;
    }
    // =================================================================================================================
    // S waves
    // =================================================================================================================
    const int fw = wv;                                 // field of this S wave: fx^2, fy^2, fx fy, fx ft, fy ft
    // derivative types of the field's two factors: 0 = fx, 1 = fy, 2 = ft.  One image's 2 x 2 filter response in the
    // reference's order (lk_device.h f2d_*): D = ((-p00 + c1 p01) + c2 p10) + c3 p11, the derivative = D(image 1) + sg D(image 2):
    //   fx: c = (+, -, +), sg = +;   fy: c = (-, +, +), sg = +;   ft: c = (-, -, -), sg = -  (f2d_pos(b) == -f2d_neg(b) exactly)
    const int T1 = (fw == 1 || fw == 4) ? 1 : 0;                       // xx, xy, xt: fx;  yy, yt: fy
    const int T2 = fw == 0 ? 0 : (fw <= 2 ? 1 : 2);                    // xx: fx;  yy, xy: fy;  xt, yt: ft
    const double c11 = T1 == 0 ? 1.0 : -1.0, c12 = T1 == 1 ? 1.0 : -1.0, c13 = T1 == 2 ? -1.0 : 1.0, sg1 = T1 == 2 ? -1.0 : 1.0;
    const double c21 = T2 == 0 ? 1.0 : -1.0, c22 = T2 == 1 ? 1.0 : -1.0, c23 = T2 == 2 ? -1.0 : 1.0, sg2 = T2 == 2 ? -1.0 : 1.0;
    double *pa_row = &s_pa[fw][PAD + lane], *pb_row = &s_pb[fw][PAD + lane], *pq_row = &s_pq[fw][PAD + lane];
    const int xs = x0 - 2 + 2 * lane;                  // image column of the lane's first column (even: the pair is all in or all out)
    const bool col_in = xs >= 0 && xs < W;
    const bool all_col_in = __builtin_amdgcn_ballot_w64(!col_in) == 0;
    const bool s_out = 2 * lane >= HL + 2 && 2 * lane < HL + 2 + XO;   // the pair is an output pair of the workgroup
    double *sums_row = &s_sums[0][fw][min(max(2 * lane - (HL + 2), 0), XO - 2)];
    double ring[L][2], pre[2] = {0.0, 0.0};
#pragma unroll
    for (int q = 0; q < L; ++q) ring[q][0] = ring[q][1] = 0.0;
    // the previous marched row: left tap, column a, column b of image 1 (A) and image 2 (B)
    double pAl = 0.0, pAa = 0.0, pAb = 0.0, pBl = 0.0, pBa = 0.0, pBb = 0.0;

    auto dpp_left = [](double v) __attribute__((always_inline)) {   // lane l <- lane l - 1 (lane 0: 0)
        const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x138, 0xF, 0xF, true);
        const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x138, 0xF, 0xF, true);
        return __hiloint2double(hi, lo);
    };
    // the two products of marched row RR of batch n (image rows from LDS, previous row from registers)
    auto sprod = [&](auto rrc, int n, double &pa, double &pb) __attribute__((always_inline)) {
        constexpr int RR = decltype(rrc)::value;
        const int i = 8 * n + RR, yy = ym0 + i;
        const bool row_ok = yy >= 0 && yy < H && i >= 1;       // (wave-uniform) row 0 of the march has no previous row
        const double *im = &s_im[n & 1][RR][0][2 * lane];
        const lk_d2 mA = *reinterpret_cast<const lk_d2 *>(im);
        const lk_d2 mB = *reinterpret_cast<const lk_d2 *>(im + 128);
        const double cAl = dpp_left(mA.y), cBl = dpp_left(mB.y);   // left tap of column a: the left lane's column b
        auto resp = [&](double k1, double k2, double k3, double p00, double p01, double p10, double p11) __attribute__((always_inline)) {
            return __builtin_fma(k3, p11, __builtin_fma(k2, p10, __builtin_fma(k1, p01, -p00)));
        };
        const double g1a = __builtin_fma(sg1, resp(c11, c12, c13, pBl, pBa, cBl, mB.x), resp(c11, c12, c13, pAl, pAa, cAl, mA.x));
        const double g1b = __builtin_fma(sg1, resp(c11, c12, c13, pBa, pBb, mB.x, mB.y), resp(c11, c12, c13, pAa, pAb, mA.x, mA.y));
        const double g2a = __builtin_fma(sg2, resp(c21, c22, c23, pBl, pBa, cBl, mB.x), resp(c21, c22, c23, pAl, pAa, cAl, mA.x));
        const double g2b = __builtin_fma(sg2, resp(c21, c22, c23, pBa, pBb, mB.x, mB.y), resp(c21, c22, c23, pAa, pAb, mA.x, mA.y));
        pAl = cAl; pAa = mA.x; pAb = mA.y; pBl = cBl; pBa = mB.x; pBb = mB.y;
        pa = g1a * g2a; pb = g1b * g2b;
        if (!(row_ok && all_col_in)) {                  // rows / columns outside the image contribute 0 (wave-uniform test)
            const bool keep = row_ok && col_in;
            pa = keep ? pa : 0.0; pb = keep ? pb : 0.0;
        }
    };
    struct HRow { double q[NQ], sb, sa; };
    // (the fences are compiler-only: seen from ONE thread the stores and the neighbours' loads do not alias, so nothing
    // else keeps the compiler from moving a load above the store of the lane it reads from; the LDS itself executes a
    // wave's instructions in order, so neither a barrier nor a wait is needed, and row r + 1 may overwrite the private row
    // as soon as the reads of row r have been ISSUED)
    auto hpublish = [&](double pa, double pb) __attribute__((always_inline)) {
        BT_WAVE_FENCE();
        pa_row[0] = pa; pb_row[0] = pb; pq_row[0] = pa + pb;
        BT_WAVE_FENCE();
    };
    auto hgather = [&](HRow &h) __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < NQ; ++d) h.q[d] = pq_row[d - QH];
        h.sb = pb_row[-HS]; h.sa = pa_row[HS];
        BT_WAVE_FENCE();
    };
    // horizontal windows of the pair + vertical van Herk step at static chunk position KC -> sums row RR
    auto svert = [&](auto kcc, auto rrc, const HRow &h) __attribute__((always_inline)) {
        constexpr int KC = decltype(kcc)::value, RR = decltype(rrc)::value;
        double hs[2];
        if constexpr (C::RODD) {
            const double shared = split_tree<0, NQ - 1>(h.q);
            hs[0] = h.sb + shared; hs[1] = shared + h.sa;
        } else {
            const double shared = split_tree<1, NQ - 2>(h.q);
            hs[0] = (h.q[0] + shared) + h.sa; hs[1] = (h.sb + shared) + h.q[NQ - 1];
        }
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
        if (s_out) *reinterpret_cast<lk_d2 *>(sums_row + RR * (5 * XO)) = o;
    };
    // two marched rows per LDS round trip: both rows' products, then publish / gather row 0 and row 1 back to back
    // (one private row: the LDS executes them in order), then both rows' sums
    auto srow2 = [&](auto kcc, auto rrc, int n) __attribute__((always_inline)) {
        constexpr int KC = decltype(kcc)::value, RR = decltype(rrc)::value;
        double pa0, pb0, pa1, pb1;
        sprod(std::integral_constant<int, RR>{}, n, pa0, pb0);
        sprod(std::integral_constant<int, RR + 1>{}, n, pa1, pb1);
        HRow h0, h1;
        hpublish(pa0, pb0); hgather(h0);
        hpublish(pa1, pb1); hgather(h1);
        svert(std::integral_constant<int, KC>{}, std::integral_constant<int, RR>{}, h0);
        svert(std::integral_constant<int, (KC + 1) % L>{}, std::integral_constant<int, RR + 1>{}, h1);
    };
    auto sbatch = [&](auto vc, int n) __attribute__((always_inline)) {
        constexpr int V = decltype(vc)::value;
        [&]<int... Ts>(std::integer_sequence<int, Ts...>) __attribute__((always_inline)) {
            (srow2(std::integral_constant<int, (8 * V + 2 * Ts) % L>{}, std::integral_constant<int, 2 * Ts>{}, n), ...);
        }(std::make_integer_sequence<int, 4>{});
    };

    constexpr int LCM = 8 * C::NV;
    int v = (((ym0 % LCM) + LCM) % LCM) >> 3;          // variant of batch 0: chunk position of marched row 0 = 8 v mod L
    __syncthreads();                                   // rows of batch 0 complete (W waves' prologue)
    for (int n = 0; n < NBS; ++n) {
        [&]<int... Vs>(std::integer_sequence<int, Vs...>) __attribute__((always_inline)) {
            ((v == Vs ? (sbatch(std::integral_constant<int, Vs>{}, n), 0) : 0), ...);
        }(std::make_integer_sequence<int, C::NV>{});
        v = v + 1 == C::NV ? 0 : v + 1;
        __syncthreads();                               // sums of batch n and rows of batch n + 1 complete
        bphase(n);
        __syncthreads();                               // sums consumed
    }
}

}  // namespace lk
