// lk_split.h -- single-level dense LK, row-marching, WAVE-SPECIALISED: the row pipeline is cut into three roles
// and the five window-sum fields are split across the waves of a block (ParallelVideoPyr.cpp:114-191, 327-542 --
// same arithmetic as lk_march3.h).
//
// Why: the one-role kernel (lk_march3.h) keeps the (WIN-1) x 5 vertical ring of a column in ONE wave's
// registers (140 of 256 VGPRs at 15 x 15 -> 2 waves per SIMD) and sums every horizontal window out of LDS
// with R+1 reads per pixel and field; a wave issues at most one instruction of any kind per ~5 cycles, so its
// ~410 instructions per row are a serial chain of its own, with one other wave per SIMD to fill the gaps.
// Here a block of 126 columns is marched by 8 waves in three roles, each with a short instruction stream:
//
//   * one "F" (front) wave, TWO adjacent columns per lane: image rows (ONE 16-byte load of the pair per image
//     and row, PF rows in flight, previous row in registers; the left derivative tap is the left lane's second
//     pixel by a DPP wave shift -- lane 0 is a halo pair, so the wave feeds 126 columns) -> fx, fy, ft -> the
//     five products of both columns (one ds_write_b128 per field).  Runs at raised wave priority: it is the one
//     wave that feeds the block from memory, and it has the longest instruction stream of the three roles.
//   * 5 "S" waves, one FIELD each, two adjacent columns per lane: both horizontal windows of the pair are the
//     R pair sums they share plus one single product each.  Round 3: a lane reads only its OWN pair (one 16-byte LDS
//     read), forms its pair sum and receives the neighbours' by DPP wave shifts (LK_SPLIT_QDPP; bit-identical to the
//     round-2 form with R 16-byte reads and R additions per row; 29 x 29 and 31 x 31 -- one workgroup per CU -- read the
//     pair sums F publishes instead, SplitCfg::QSUM); + 2 eight-byte reads of the single products; then the vertical
//     van Herk / Gil-Werman ring for the two columns in registers ((WIN-1) x 2 x 2 = 56 VGPRs at 15 x 15), chunks
//     anchored to absolute image rows (results do not depend on the strip height); the two window sums go back through
//     LDS (one 16-byte store).
//   * 2 "B" (back) waves, one column per lane: the five window sums of a row come back from LDS (one
//     ds_read_b64 per field) -> 2 x 2 solve with true fp64 divisions, /0 -> 0, "+= pyrUp(coarse flow)" from
//     register-carried coarse rows (taps fetched one step ahead), streaming store.  The frame of width R just inside
//     the image (clipped windows, outermost ring) is never stored by B: lk_border_pixel writes it, in trailing
//     workgroups of the same launch.
//
// The three stages run concurrently on different marched rows with double-buffered LDS planes; a pipeline STEP is
// TWO marched rows (F: rows 2s, 2s+1; S: the two rows of step s-1; B: those of step s-2) and ends in ONE workgroup
// barrier: the LDS round trips, the barrier and the dependent fp64 chains of the solve are paid once per two rows,
// and a B wave interleaves the two independent solves of its row pair.  Every vector-memory instruction is
// issued unconditionally (clamped rows; a store that must not happen gets an out-of-range buffer offset, which
// the hardware drops): hipcc's s_waitcnt counting gives up on loads or stores under a condition and then drains
// vmcnt(0) at every use, which serialised a step behind its own prefetch (F) and behind its own stores (B).
// <= 128 VGPRs up to 27 x 27 (round 3; 21 x 21 before the DPP pair sums freed the S waves' read buffers): four waves per
// SIMD = TWO blocks per CU, whose phases interleave (one block per CU marches in lock-step: every wave reads LDS,
// computes and writes at the same time).  1-D grid, XCD-aware tile
// order (column blocks of a strip and vertically adjacent strips share halo lines through one L2).
//
// FW = true (round 4: the default for tall levels, LKFLOW_OPT_FUSE_WARP): FOUR more "W" waves per block run the warp of the level
// (a9-a12: cv::pyrUp of the coarse flow, float map, fixed-point remap of both UN-warped images; wave (half, e) makes row e of a
// step's pair for 64 of the 128 columns) and hand the rows to F through LDS, so the warped planes never reach memory; F then has
// no global loads of its own and serves W the horizontally filtered coarse rows.  12 waves = three per SIMD, <= 168 VGPRs, 89 KB
// LDS: one workgroup per CU.  Gathers are issued two pipeline steps before their blend (two tap buffers), the four float weights
// of a sample come from a 32 KB LDS table, frame pixels are made in trailing workgroups (lk_border_pixel<., true>).  Bit-identical
// to warp_kernel + the unfused kernel; 0.94 ms against 0.52 + 0.56 per 32 x 1080p at 15 x 15 (DESIGN.md sections 4.3, 4.5; round 2's
// two-wave form was slower than warp + LK: section 4.4).
//
// Only additions in the window sums (flat regions give exactly 0 like the reference); same products, solve and pyrUp
// expressions as lk_march3.h.  Host-checked: W and H even below a coarser level (a coarsest level may be odd: see nrows and half_fix),
// H > R + PF + 1, even strip height (a step's row pair is an even and an odd output row); other shapes take lk_march3.h.  The LK_ABL_* macros are timing-only ablations (results wrong by
// design) behind the records in profiles/r04_a_fused_warp_ab.txt; no shipped build defines them.
#pragma once
#include "lk_w3x2.h"     // buf_ld2 (16-byte buffer loads), lk_march3.h

#ifndef LK_SPLIT_PF
#define LK_SPLIT_PF 2    /* image rows an F lane keeps in flight (2, 4, 8: no measurable difference) */
#endif
#ifndef LK_SPLIT_PRIO_F
#define LK_SPLIT_PRIO_F 3
#endif
#ifndef LK_SPLIT_PRIO_W
#define LK_SPLIT_PRIO_W 0   /* (2 gained 3 % while the W waves waited for their gathers at the start of a step; with the gathers two steps ahead 0 is 3.5 % faster than 1 .. 3) */
#endif
#ifndef LK_SPLIT_BOTH_MAX
#define LK_SPLIT_BOTH_MAX 15   /* widest window whose S waves issue the LDS reads of both rows of a step up front */
#endif
#ifndef LK_SPLIT_FW_NOQDPP
#define LK_SPLIT_FW_NOQDPP 0
#endif
#ifndef LK_SPLIT_QDPP
#define LK_SPLIT_QDPP 1  /* S waves up to 21 x 21: neighbours' pair sums by DPP wave shifts instead of LDS reads (A/B: profiles/r03_c_*) */
#endif
#ifndef LK_SPLIT_TWO_WG_MAX
#define LK_SPLIT_TWO_WG_MAX 27   /* widest window compiled for 128 VGPRs = two 8-wave workgroups per CU */
#endif
#ifndef LK_SPLIT_QSUM_ABOVE
#define LK_SPLIT_QSUM_ABOVE 21   /* windows above this: F publishes the pair sums (SplitCfg::QSUM) */
#endif
#ifndef LK_SPLIT_TD
#define LK_SPLIT_TD 2    /* even: row pairs of coarse taps a B lane keeps in flight (4 spills at 15x15 and is slower) */
#endif



namespace lk {

typedef double lk_d2 __attribute__((ext_vector_type(2)));

template <int WIN, int K, bool FW = false>
struct SplitCfg {
    static constexpr int R = WIN / 2;
    static constexpr int NWW = FW ? 2 * K : 0;                // W waves (fused warp): 64 columns and ONE row of the step's pair each, one column per lane
    static constexpr int LAG = FW ? 2 : 0;           // pipeline steps the W role runs ahead of F
    static constexpr int PF = LK_SPLIT_PF;           // image rows in flight per F lane: a pipeline step is short, HBM latency is not
    static constexpr int NB = K;                     // B waves: 64 columns each
#ifndef LK_SPLIT_NF2_MIN
#define LK_SPLIT_NF2_MIN 23   /* windows LK_SPLIT_NF2_MIN .. LK_SPLIT_NF2_MAX: two F waves per workgroup */
#endif
#ifndef LK_SPLIT_NF2_MAX
#define LK_SPLIT_NF2_MAX 27
#endif
    // F waves: lane 0 is a halo pair (left tap of lane 1), lanes 1..63 = 126 columns.  Windows that run ONE workgroup per
    // CU anyway (ring > 128 VGPRs) have room for a ninth wave, and there the F instruction stream paces the workgroup
    // (F publishing quad sums as well, +30 instructions per step, cost +13 % at 31 x 31): two F waves march the same rows
    // (same loads -- the second copy hits the CU's L1) and each computes the derivatives and products of ONE row of the
    // step's pair.  Bit-identical.  Nine waves put three on one SIMD = 168 VGPRs: 23 x 23 .. 27 x 27 fit (143 .. 165) and
    // gain 4-5 % (3840 x 2160 / L5, level-0 LK 1.00 -> 0.95 ms at 23 x 23, 1.03 -> 0.995 at 27 x 27); the 29 x 29 and
    // 31 x 31 rings would spill (31 x 31: 68 bytes, 8K/L5/w31 level-0 LK 2.02 -> 2.45 ms) and keep one F wave
    // (profiles/r02_j_bigwin_knobs_ab.txt).
#ifndef LK_SPLIT_FW_NF
#define LK_SPLIT_FW_NF 1   /* fused warp: F waves per workgroup (2: each makes the products of ONE row of the step's pair -- 13 waves put four on one SIMD: measured 1.08 -> 1.37 ms, profiles/r04_a_fused_warp_ab.txt) */
#endif
    static constexpr int NF = FW ? LK_SPLIT_FW_NF : ((WIN >= LK_SPLIT_NF2_MIN && WIN <= LK_SPLIT_NF2_MAX && WIN > LK_SPLIT_TWO_WG_MAX) ? 2 : 1);
    static constexpr int TXC = 64 * K - 2;           // columns per block with products
    static constexpr int HL = R + (R & 1);           // halo columns on each side (even: a lane's column pair is 16-byte aligned)
    static constexpr int XO = TXC - 2 * HL;          // output columns per block
    static constexpr int NW = NB + 5 + NF + NWW;     // waves per block, in this order: B, S, F, W
    static constexpr int NT = 64 * NW;
    static constexpr int KH = K / 2;                 // 16-byte chunks (column pairs) per S lane
    static constexpr int PADJ = (R + K - 1) / K;     // pad lanes on each side of a product plane
    static constexpr int NJP = 64 + 2 * PADJ;        // chunks per product half-plane
    // half-plane strides in 16-byte chunks: consecutive F lanes write alternate halves (16-byte stores, banks
    // mod 32 -> the halves sit 16 banks apart); B lanes read sums 8 bytes per lane (banks mod 64 -> 32 apart)
    static constexpr int PSTR = KH == 1 ? NJP : ((NJP + 3) / 8 * 8 + 4);
    static constexpr int SSTR = KH == 1 ? 64 : 72;
    static constexpr int PROD_D = KH * PSTR * 2;     // doubles per product plane
    static constexpr int SUMS_D = KH * SSTR * 2;     // doubles per sums plane
    // waves per SIMD to compile for (8-wave blocks: 4 -> two blocks per CU; 10-wave blocks with the W role: 5 -> two blocks)
    static constexpr int WPE = FW ? (NW + 3) / 4 : ((K == 2 && WIN <= LK_SPLIT_TWO_WG_MAX) ? 4 : (NF == 2 ? 3 : 1));   // nine waves: three on one SIMD
    // F publishes the pair sums a + b (true) or S adds the raw pairs itself (false).  Up to 21 x 21 (two workgroups per CU)
    // F is the wave that paces a workgroup and S has slack: S adds (4K/L5/w21 28.4 -> 30.2 Gpx/s).  From 23 x 23 on (one
    // workgroup per CU, 16 LDS reads per row and field) the S waves' LDS reads are the longer pole: halving their bytes
    // wins (8K/L5/w31: level-0 LK 2.20 -> 2.08 ms; profiles/r02_j_qsum_ab.txt).
    static constexpr bool QSUM = !FW && WIN > LK_SPLIT_QSUM_ABOVE && WIN > LK_SPLIT_TWO_WG_MAX;   // (fused warp: S sums as the shifted pair sums arrive, at any window)
    static constexpr int UN = PF >= 4 ? PF / 2 : 2;  // unrolled F group in steps: even (LDS buffer parity) and PF rows (queue slots)
    static_assert(K == 2, "one F wave feeds 126 columns: S lanes own 2 columns (K = 4 ran one 11-wave block per CU: slower)");
    static_assert(PSTR >= NJP, "product half-planes must not overlap");
    static_assert(PF == 2 || (PF % 4 == 0 && PF <= 16), "static queue slots: two rows per step, an even number of steps per group");
};

// lane l <- lane l - 1 / lane l + 1 of the wave (edge lanes: 0), without the zero-initialised "old" operand of dpp_wave_shr1
__device__ __forceinline__ double dpp_from_left(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x138, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x138, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_from_right(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x130, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x130, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// tree sum of v[LO..HI] (compile-time bounds)
template <int LO, int HI, typename A>
__device__ __forceinline__ double split_tree(const A &v)
{
    if constexpr (LO == HI) return v[LO];
    else {
        constexpr int MID = LO + (HI - LO) / 2;
        return split_tree<LO, MID>(v) + split_tree<MID + 1, HI>(v);
    }
}

// ------------------------------------------------------------------------------------------------
// Clipped-window pixels of lk_split_kernel's output (SerialPyr.cpp:94-95: a pixel inside the outermost ring whose
// WIN x WIN window leaves the image takes m(y,x) * WIN^2 for each of the five sums): the frame of width R just
// inside the image INCLUDING the outermost ring (0 + pyrUp), one thread per pixel; the marching workgroups store none of
// it, so there is no ordering between the two.  Runs as trailing workgroups of lk_split_kernel's own launch (default) or
// as lk_border_kernel.  Same expressions as
// the one-role kernels' clipped branch (lk_products_at, lk_solve, pyrup_at).
// ------------------------------------------------------------------------------------------------
template <bool HAS_UP, bool FW = false>
__device__ __forceinline__ void lk_border_pixel(const LKArgs &a, int win, unsigned idx, int pair)
{
    // (host-checked: the frame of one plane has fewer than 2^31 pixels -- 32-bit index arithmetic; a 64-bit division
    // costs ~100 vector instructions and this function ran three of them)
    const int R = win / 2, H = a.H, W = a.W;
    int y, x;
    const unsigned nr = (unsigned)min(2 * R, H);             // rows of the top + bottom bands
    if (idx < nr * (unsigned)W) {
        const unsigned k = idx / (unsigned)W;
        x = (int)(idx - k * (unsigned)W);
        y = (H <= 2 * R) ? (int)k : ((int)k < R ? (int)k : H - 2 * R + (int)k);
    } else {                                                 // left + right bands of the rows in between
        const unsigned k2 = idx - nr * (unsigned)W;
        const unsigned nc = (unsigned)min(2 * R, W);
        if (H <= 2 * R || k2 >= (unsigned)(H - 2 * R) * nc) return;
        const unsigned q = k2 / nc;
        y = R + (int)q;
        const int c = (int)(k2 - q * nc);
        x = (W <= 2 * R) ? c : (c < R ? c : W - 2 * R + c);
    }
    if (!((y - R < 0) || (y + R > H - 1) || (x - R < 0) || (x + R > W - 1))) return;      // full window: the main kernel's (never mapped here)
    const bool ring = !(y >= 1 && y <= H - 2 && x >= 1 && x <= W - 2);                    // outermost ring: 0 (+ pyrUp)
    double S[5], u = 0.0, v = 0.0;
    if constexpr (FW) {
        // Fused warp: a.i1 / a.i2 are the UN-warped level images, so the four derivative taps (rows yc - 1, yc x columns
        // xc - 1, xc, all inside the image for a pixel inside the ring; ring pixels compute a neighbour's and drop it) of
        // both images are MADE here: cv::pyrUp of the coarse flow at the four positions from ONE 3-row x 6-tap patch per
        // flow plane (rows s(i-1), s(i), s(i+1), i = yc >> 1, cover both rows; make_upsel1 / pyrup_h1v: the main kernel's
        // roundings), float map, fixed-point remap with all sixteen gathers issued unconditionally and together
        // (lk_device.h remap_taps / remap_blend: warp_kernel's bits).  Straight-line, no load under a condition.
        const int yc = max(y, 1), xc = max(x, 1);
        const int Hc = H >> 1, Wc = W >> 1, i = yc >> 1;
        const UpSel1 usc[2] = {make_upsel1(xc - 1, Wc), make_upsel1(xc, Wc)};
        const double *cu = a.cu.at(pair), *cv = a.cv.at(pair);
        double tu[3][2][3], tv[3][2][3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int k = i - 1 + q;
            const int sr = k < 0 ? min(1, Hc - 1) : (k > Hc - 1 ? Hc - 1 : k);
            const char *pu = reinterpret_cast<const char *>(cu + (long long)sr * a.cu.pitch);
            const char *pv = reinterpret_cast<const char *>(cv + (long long)sr * a.cv.pitch);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                tu[q][c][0] = *reinterpret_cast<const double *>(pu + usc[c].jm); tu[q][c][1] = *reinterpret_cast<const double *>(pu + usc[c].jc);
                tu[q][c][2] = *reinterpret_cast<const double *>(pu + usc[c].jp);
                tv[q][c][0] = *reinterpret_cast<const double *>(pv + usc[c].jm); tv[q][c][1] = *reinterpret_cast<const double *>(pv + usc[c].jc);
                tv[q][c][2] = *reinterpret_cast<const double *>(pv + usc[c].jp);
            }
        }
        double upu[2][2], upv[2][2];                                   // [row yc - 1, yc][column xc - 1, xc]
        const bool yodd = (yc & 1) != 0;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            double hu[3], hv[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                hu[q] = pyrup_h1v(tu[q][c][0], tu[q][c][1], tu[q][c][2], usc[c]);
                hv[q] = pyrup_h1v(tv[q][c][0], tv[q][c][1], tv[q][c][2], usc[c]);
            }
            const double ue = (hu[0] + hu[1] * 6 + hu[2]) * (1.0 / 64), ve = (hv[0] + hv[1] * 6 + hv[2]) * (1.0 / 64);   // the even row of the two
            // the odd row: below the even one (yc odd: rows s(i), s(i+1)) or above it (yc even: rows s(i-1), s(i))
            const double uo = (((yodd ? hu[1] : hu[0]) + (yodd ? hu[2] : hu[1])) * 4) * (1.0 / 64);
            const double vo = (((yodd ? hv[1] : hv[0]) + (yodd ? hv[2] : hv[1])) * 4) * (1.0 / 64);
            upu[0][c] = yodd ? ue : uo; upu[1][c] = yodd ? uo : ue;
            upv[0][c] = yodd ? ve : vo; upv[1][c] = yodd ? vo : ve;
        }
        const double *I1 = a.i1.at(pair), *I2 = a.i2.at(pair);
        const int ip = (int)a.i1.pitch;
        RemapTaps t1[2][2], t2[2][2];
        lk_d2u g1[2][2][2], g2[2][2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float xf = (float)(xc - 1 + c), yf = (float)(yc - 1 + r);
                const float hx = (float)upu[r][c] / 2, hy = (float)upv[r][c] / 2;      // map build (ParallelVideoPyr.cpp:674-678), float32
                t1[r][c] = remap_taps(ip, H, W, xf + hx, yf + hy);
                t2[r][c] = remap_taps(ip, H, W, xf - hx, yf - hy);
                g1[r][c][0] = *reinterpret_cast<const lk_d2u *>(I1 + t1[r][c].off); g1[r][c][1] = *reinterpret_cast<const lk_d2u *>(I1 + t1[r][c].off + ip);
                g2[r][c][0] = *reinterpret_cast<const lk_d2u *>(I2 + t2[r][c].off); g2[r][c][1] = *reinterpret_cast<const lk_d2u *>(I2 + t2[r][c].off + ip);
            }
        double wa[2][2], wb[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                wa[r][c] = remap_blend(t1[r][c], g1[r][c][0], g1[r][c][1], H, W);
                wb[r][c] = remap_blend(t2[r][c], g2[r][c][0], g2[r][c][1], H, W);
            }
        lk_products(wa[0][0], wa[0][1], wa[1][0], wa[1][1], wb[0][0], wb[0][1], wb[1][0], wb[1][1], S);
        for (int f = 0; f < 5; ++f) S[f] = S[f] * (double)(win * win);
        double us_, vs_;
        lk_solve(S, us_, vs_);
        // "+= pyrUp" at (y, x) itself: one of the four positions (a ring pixel sits on row yc - 1 and / or column xc - 1)
        const int rr = y == yc ? 1 : 0, cc = x == xc ? 1 : 0;
        u = (ring ? 0.0 : us_) + (rr ? (cc ? upu[1][1] : upu[1][0]) : (cc ? upu[0][1] : upu[0][0]));
        v = (ring ? 0.0 : vs_) + (rr ? (cc ? upv[1][1] : upv[1][0]) : (cc ? upv[0][1] : upv[0][0]));
        a.u.at(pair)[(long long)y * a.u.pitch + x] = u;
        a.v.at(pair)[(long long)y * a.v.pitch + x] = v;
        return;
    } else {
        // Straight-line: every load is issued up front from an address that is valid for ring pixels too (their products
        // are computed and dropped), so that the image taps and the coarse-flow taps are in flight together.  A pixel
        // inside the ring has y - 1 >= 0 and x - 1 >= 0: no reflection.
        const int yc = max(y, 1), xc = max(x, 1);
        const double *r1a = a.i1.at(pair) + (long long)yc * a.i1.pitch + xc, *r0a = r1a - a.i1.pitch;
        const double *r1b = a.i2.at(pair) + (long long)yc * a.i2.pitch + xc, *r0b = r1b - a.i2.pitch;
        const double a00 = r0a[-1], a01 = r0a[0], a10 = r1a[-1], a11 = r1a[0];
        const double b00 = r0b[-1], b01 = r0b[0], b10 = r1b[-1], b11 = r1b[0];
        [[maybe_unused]] double tu[3][3], tv[3][3];
        [[maybe_unused]] UpSel1 us{};
        if constexpr (HAS_UP) {
            // cv::pyrUp of the coarse flow at (y, x) with the main kernel's branch-free taps (make_upsel1 / pyrup_h1v:
            // the same roundings as pyrup_h) on the coarse rows s(i-1), s(i), s(i+1), i = y >> 1 (clamped as pyrup_at's
            // reflections come out: s(-1) = min(1, Hc-1), s(Hc) = Hc-1)
            const int Hc = H >> 1, Wc = W >> 1, i = y >> 1;
            us = make_upsel1(x, Wc);
            const double *cu = a.cu.at(pair), *cv = a.cv.at(pair);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int k = i - 1 + q;
                const int sr = k < 0 ? min(1, Hc - 1) : (k > Hc - 1 ? Hc - 1 : k);
                const char *pu = reinterpret_cast<const char *>(cu + (long long)sr * a.cu.pitch);
                const char *pv = reinterpret_cast<const char *>(cv + (long long)sr * a.cv.pitch);
                tu[q][0] = *reinterpret_cast<const double *>(pu + us.jm); tu[q][1] = *reinterpret_cast<const double *>(pu + us.jc);
                tu[q][2] = *reinterpret_cast<const double *>(pu + us.jp);
                tv[q][0] = *reinterpret_cast<const double *>(pv + us.jm); tv[q][1] = *reinterpret_cast<const double *>(pv + us.jc);
                tv[q][2] = *reinterpret_cast<const double *>(pv + us.jp);
            }
        }
        lk_products(a00, a01, a10, a11, b00, b01, b10, b11, S);
        for (int f = 0; f < 5; ++f) S[f] = S[f] * (double)(win * win);
        double us_, vs_;
        lk_solve(S, us_, vs_);
        u = ring ? 0.0 : us_;
        v = ring ? 0.0 : vs_;
        if constexpr (HAS_UP) {
            double hu[3], hv[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                hu[q] = pyrup_h1v(tu[q][0], tu[q][1], tu[q][2], us);
                hv[q] = pyrup_h1v(tv[q][0], tv[q][1], tv[q][2], us);
            }
            const bool odd = (y & 1) != 0;
            const double ue = (hu[0] + hu[1] * 6 + hu[2]) * (1.0 / 64), uo = ((hu[1] + hu[2]) * 4) * (1.0 / 64);
            const double ve = (hv[0] + hv[1] * 6 + hv[2]) * (1.0 / 64), vo = ((hv[1] + hv[2]) * 4) * (1.0 / 64);
            u += odd ? uo : ue;
            v += odd ? vo : ve;
        }
        a.u.at(pair)[(long long)y * a.u.pitch + x] = u;
        a.v.at(pair)[(long long)y * a.v.pitch + x] = v;
        return;
    }
    if (HAS_UP) {
        u += pyrup_at(a.cu.at(pair), a.cu.pitch, H >> 1, W >> 1, y, x);
        v += pyrup_at(a.cv.at(pair), a.cv.pitch, H >> 1, W >> 1, y, x);
    }
    a.u.at(pair)[(long long)y * a.u.pitch + x] = u;
    a.v.at(pair)[(long long)y * a.v.pitch + x] = v;
}

// the same as a launch of its own (LKFLOW_BORDER_MERGED=0, A/B): grid (frame pixels / 256, pairs)
template <bool HAS_UP, bool FW = false>
__global__ __launch_bounds__(256) void lk_border_kernel(LKArgs a, int win)
{
    lk_border_pixel<HAS_UP, FW>(a, win, blockIdx.x * 256u + threadIdx.x, blockIdx.y);
}

template <int WIN, int K, bool HAS_UP, bool FW = false>
__global__ __launch_bounds__((SplitCfg<WIN, K, FW>::NT), (SplitCfg<WIN, K, FW>::WPE)) void lk_split_kernel(LKArgs a)
{
    using C = SplitCfg<WIN, K, FW>;
    static_assert(!FW || HAS_UP, "the fused warp needs a coarser level");
    constexpr int R = C::R, XO = C::XO, KH = C::KH, PADJ = C::PADJ, PF = C::PF, UN = C::UN, HL = C::HL;
    if (a.nmain && blockIdx.x >= a.nmain) {
        // trailing workgroups of the launch: the frame of width R just inside the image (lk_border_pixel above), one
        // pixel per thread.  They are dispatched last, into the slots the marching workgroups free as they finish, so
        // the frame costs neither a launch of its own nor the idle tail of one.
        const unsigned bb = blockIdx.x - a.nmain;
        lk_border_pixel<HAS_UP, FW>(a, WIN, (bb % a.nbb) * (unsigned)C::NT + threadIdx.x, (int)(bb / a.nbb));
        return;
    }
    __shared__ __attribute__((aligned(16))) double s_prod[2][2][5][C::PROD_D];   // [step parity][row of the step][field]: (a, b) products of every column pair
    __shared__ double s_q[C::QSUM ? 2 : 1][C::QSUM ? 2 : 1][C::QSUM ? 5 : 1][C::QSUM ? C::NJP : 1];   // the pair sums a + b (index = chunk), if F publishes them
    __shared__ __attribute__((aligned(16))) double s_sums[2][2][5][C::SUMS_D];
    // FW: the warped rows W hands to F: [step parity][row of the step][image][block column - (-2)] (128 columns: F's 64 pairs)
    __shared__ __attribute__((aligned(16))) double s_w[FW ? 2 : 1][FW ? 2 : 1][FW ? 2 : 1][FW ? 128 : 2];
    // FW: F's coarse-row service: [step parity][u, v][block column + 2]: the horizontally filtered coarse row W rotates in
    __shared__ __attribute__((aligned(16))) double s_c[FW ? 2 : 1][FW ? 2 : 1][FW ? 128 : 2];
    // FW: cv::remap's bilinear weight table (OpenCV 2.4 keeps the same table, imgwarp.cpp initInterTab2D): entry (sy & 31) * 32 +
    // (sx & 31) = the four float weights (1-b)(1-a), (1-b)a, b(1-a), ba as doubles (each an exact multiple of 1/1024) -- two
    // 16-byte LDS reads per sample instead of 16 vector instructions (and / cvt / mul / sub / 4 mul / 4 cvt)
    __shared__ __attribute__((aligned(16))) double s_wt[FW ? 1024 : 1][FW ? 4 : 1];

    // logical wave (role order B, S, F, W) of the physical wave.  A workgroup's waves are dealt to the CU's four SIMDs cyclically
    // (physical waves p, p + 4, p + 8 ... share one), so for the fused kernel's 12 or 13 waves a table (-DLK_SPLIT_FW_PERM, one hex
    // digit per physical wave, wave 0 lowest) chooses which roles share a SIMD; 0 = role order
#ifndef LK_SPLIT_FW_PERM
#define LK_SPLIT_FW_PERM 0
#endif
    const int pwv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int wv = (FW && LK_SPLIT_FW_PERM != 0) ? (int)(((unsigned long long)LK_SPLIT_FW_PERM >> (4 * pwv)) & 15) : pwv;
    const int ltid = wv * 64 + ((int)threadIdx.x & 63);   // thread index in role order
    // 1-D grid, XCD-aware tile order (lk_kernels.h): workgroups are dealt round-robin over the 8 XCDs (private L2s), so
    // with the natural order the column blocks of one strip -- which share 2 HL halo columns -- and vertically adjacent
    // strips -- which share 2R apron rows -- land on different L2s and every shared line is fetched from HBM once per XCD
    const SplitTile st = split_tile(a, blockIdx.x, a.nmain ? a.nmain : gridDim.x);
    const int bx = st.bx, pair = st.pair;
    const int H = a.H, W = a.W;
    const int y0 = st.y0;                            // even (host)
    const int y1 = min(y0 + st.th, H);
    const int nrows = y1 - y0;                       // even, except in the last strip of an odd level (host: only a coarsest level, !HAS_UP): its
                                                     // last step marches the row below the image as well -- zero products (F), store dropped (B)
    const int nm = nrows + 2 * R;                    // marched product rows: image rows y0-R .. y1-1+R
    const int yy0 = y0 - R;
    const int x0 = a.xorg + bx * XO - HL;            // image column of block column 0 (even; LKArgs::xorg)
    const int nsw = (nm + 1) / 2;                    // steps in which F (and, one step later, S) has rows
    const int ns = nsw + 2 + C::LAG;                 // pipeline steps: B trails F by two (FW: F trails W by two)
    const int NS = (ns + UN - 1) / UN * UN;          // barriers per wave: padded to whole F groups (idle tail steps)

    // zero the pad lanes of the product planes once (only non-output columns ever read them)
    constexpr int PZ = PADJ + 1;                      // right side: the pads and the chunk of columns TXC, TXC + 1
    for (int k = threadIdx.x; k < 4 * 5 * KH * 2 * PZ; k += C::NT) {
        const int side = k % 2, rest = k / 2;
        const int pj = rest % PZ, h = (rest / PZ) % KH, f = (rest / (PZ * KH)) % 5, b = rest / (PZ * KH * 5);
        const int chunk = h * C::PSTR + (side ? PADJ + 63 + pj : min(pj, PADJ - 1));
        s_prod[b >> 1][b & 1][f][2 * chunk] = 0.0;
        s_prod[b >> 1][b & 1][f][2 * chunk + 1] = 0.0;
        if constexpr (C::QSUM) s_q[b >> 1][b & 1][f][chunk] = 0.0;
    }
    if constexpr (FW) {
        for (int k = threadIdx.x; k < 1024; k += C::NT) {
            const float fa = (float)(k & 31) * (1.0f / 32), fb = (float)(k >> 5) * (1.0f / 32);   // remap_bilinear's own expressions
            s_wt[k][0] = (double)((1.0f - fb) * (1.0f - fa)); s_wt[k][1] = (double)((1.0f - fb) * fa);
            s_wt[k][2] = (double)(fb * (1.0f - fa)); s_wt[k][3] = (double)(fb * fa);
        }
    }
    __syncthreads();

    // FW: coarse-row bookkeeping shared by the W role and by F's coarse-row service.  A W-step's row pair holds one even
    // row ye (slot EV; yy0 has R's parity); entering it moves the carried coarse rows s(i-1), s(i), s(i+1) on to
    // i = ye >> 1 by rotating in s((ye >> 1) + 1) -- unless ye < 2 (rows above the image use row 1's pair) or below the image
    [[maybe_unused]] constexpr int EV = R & 1;
    [[maybe_unused]] const int ye0 = yy0 - 2 + EV;                       // the even row of W-step 0
    [[maybe_unused]] const int ib = ye0 >= 2 ? (ye0 >> 1) - 1 : 0;       // the carried rows start as s(ib-1), s(ib), s(ib+1)
    [[maybe_unused]] const int c_lo = max(ib - 1, 0);                    // first coarse row the descriptors cover
    [[maybe_unused]] auto up_srow_w = [&](int k) { const int Hc = H >> 1; return k < 0 ? min(1, Hc - 1) : (k > Hc - 1 ? Hc - 1 : k); };
    [[maybe_unused]] auto rot_row_off = [&](int t) { return max(up_srow_w(((ye0 + 2 * t) >> 1) + 1) - c_lo, 0); };   // row W-step t rotates in

#ifdef LK_ABL_BARRIERS_ONLY   /* timing-only ablation: every wave only takes part in the pipeline's barriers */
    for (int k = 0; k < NS + 1 + (FW ? 1 : 0); ++k) __syncthreads();
    return;
#endif
    if (FW && wv >= C::NB + 5 + C::NF) {
      if constexpr (FW) {                            // (not even compiled for the unfused kernel)
        // =========================================================================================
        // W waves (fused warp, a9-a12): one column per lane, ONE row of the step's pair per wave -> the warped rows of both
        // images, through LDS to F
        // =========================================================================================
        // cv::pyrUp of the coarse flow (incremental: the three horizontally filtered coarse rows of the current row
        // pair are carried in registers; the new row of a step comes from F's coarse-row service through LDS), float
        // map, 1/32-pixel fixed-point bilinear remap with fp64 blend: warp_kernel's arithmetic bit for bit (lk_kernels.h,
        // lk_device.h remap_bilinear).  W-step t produces the marched rows yy0 - 2 + 2t and yy0 - 1 + 2t into plane
        // t & 1; F reads them in step t + 1 (so F-step s runs in pipeline step s + 2).  Four W waves: wave (half, e) makes
        // row e of the pair for 64 of the 128 columns (round 4; round 2 ran two waves with both rows each, whose
        // instruction stream was twice as long as any other role's and paced the workgroup).  Software pipeline: step t
        // blends the taps gathered for step t, issues the gathers of step t + 1 (addresses ready since step t - 1) and
        // then computes the addresses of step t + 2 while they fly.  Gathers are unconditional, from clamped addresses:
        // taps that fall outside the image are zeroed afterwards (BORDER_CONSTANT 0) in a wave-uniform branch that holds no
        // memory instruction.
        const int widx = wv - (C::NB + 5 + C::NF);   // 0 .. NWW-1
        if (LK_SPLIT_PRIO_W) __builtin_amdgcn_s_setprio(LK_SPLIT_PRIO_W);   // (A/B knob; see the definition)
        const int g = (widx & 1) * 64 + ((int)threadIdx.x & 63);   // block column g - 2: the two columns of F lane g >> 1
        const int Hc = H >> 1, Wc = W >> 1;
        const int xcol = min(max(x0 + 2 * ((g >> 1) - 1), 0), W - 2) + (g & 1);   // F's clamped column pair
        const float xf = (float)xcol;
        const int ipb = (int)a.i1.pitch * 8;         // (host: both images have the same pitch)
        const long long img_bytes = ((long long)(H - 1) * a.i1.pitch + W) * 8;
        const __amdgpu_buffer_rsrc_t rI1 = buf_rsrc(a.i1.at(pair), img_bytes);
        const __amdgpu_buffer_rsrc_t rI2 = buf_rsrc(a.i2.at(pair), img_bytes);
        auto wrole = [&](auto ec) __attribute__((always_inline)) {
        constexpr int E = decltype(ec)::value;       // this wave's row of every step's pair
        double hu[3], hv[3];
        struct WAddr { int vo[2], wi[2], dx[2], dy[2]; };  // per image: gather byte offset, weight-table byte offset, distance of the real tap block from the clamped one (0, 0: all taps inside)
        // rotation of W-step t with the new coarse row (nu, nv), up-sampled flow of this wave's row, map, fixed-point coordinates
        auto waddr = [&](int t, double nu, double nv, WAddr &A) __attribute__((always_inline)) {
            const int ye = ye0 + 2 * t;
            const bool rot = ye >= 2 && ye <= H - 1;
            double uu = 0.0, vv = 0.0;
            if constexpr (EV == 1 && E == 0) {                          // the odd row above the even one: before the rotation
                uu = ((hu[1] + hu[2]) * 4) * (1.0 / 64); vv = ((hv[1] + hv[2]) * 4) * (1.0 / 64);
            }
            if (rot) {
                hu[0] = hu[1]; hu[1] = hu[2]; hu[2] = nu;
                hv[0] = hv[1]; hv[1] = hv[2]; hv[2] = nv;
            }
            if constexpr (E == EV) {
                uu = (hu[0] + hu[1] * 6 + hu[2]) * (1.0 / 64); vv = (hv[0] + hv[1] * 6 + hv[2]) * (1.0 / 64);
            }
            if constexpr (EV == 0 && E == 1) {
                uu = ((hu[1] + hu[2]) * 4) * (1.0 / 64); vv = ((hv[1] + hv[2]) * 4) * (1.0 / 64);
            }
            const int yy = yy0 - 2 + 2 * t + E;
            // rows outside the image: only row -1 is ever used (filter2D's reflect-101 tap of row 0) = row 1; the
            // others feed products that F zeroes, any in-image row will do
            const float yf = (float)min(yy < 0 ? -yy : yy, H - 1);
            const float hx = (float)uu / 2, hy = (float)vv / 2;   // map build (ParallelVideoPyr.cpp:674-678), float32
#ifdef LK_ABL_W_NOADDR   /* timing-only ablation: results wrong */
            A.vo[0] = A.vo[1] = (int)(hx + hy + yf) & 0xFF8; A.wi[0] = A.wi[1] = 0; A.dx[0] = A.dx[1] = A.dy[0] = A.dy[1] = 0;
            if (0)
#endif
#pragma unroll
            for (int im = 0; im < 2; ++im) {
                const float mx = im ? xf - hx : xf + hx, my = im ? yf - hy : yf + hy;
                const int sx = cv_round_f(mx * 32.0f), sy = cv_round_f(my * 32.0f);
                // (host-checked for this kernel: W, H <= 32766, so cv::remap's saturation of the tap coordinates to short cannot move
                // a coordinate into or inside the image: clamping the unsaturated value gives the same address, and its distance
                // from the clamp classifies the taps -- 0, +-1, or anything else = all outside -- like the saturated one)
                const int ix = sx >> 5, iy = sy >> 5;
                const int ixc = min(max(ix, 0), W - 2), iyc = min(max(iy, 0), H - 2);
                // (host-checked for this kernel: the plane has fewer than 2^24 bytes per row and fewer than 2^24 rows -- 24-bit multiply)
                A.vo[im] = (int)__umul24((unsigned)iyc, (unsigned)ipb) + ixc * 8;
                A.wi[im] = ((sy & 31) << 10) | ((sx & 31) << 5);          // byte offset of the 32-byte table entry
                A.dx[im] = ix - ixc; A.dy[im] = iy - iyc;
            }
        };
        // gathers in flight: [image][tap row] = columns (ixc, ixc + 1); TWO sets, so that the gathers of step t + 1 are issued at the
        // very start of step t while the blend of step t -- at its end -- still reads the other set: a gather has two pipeline steps
        // to land (round-4 first form: half a step; HBM latency under load is longer than a step)
        lk_d2 Ga[2][2], Gb[2][2];
        auto wissue = [&](const WAddr &A, lk_d2 (&G)[2][2]) __attribute__((always_inline)) {
#pragma unroll
            for (int im = 0; im < 2; ++im) {
                const __amdgpu_buffer_rsrc_t rI = im ? rI2 : rI1;
                double g0x, g0y, g1x, g1y;
#ifdef LK_ABL_W_NOLOAD   /* timing-only ablation: results wrong */
                g0x = g0y = g1x = g1y = (double)A.vo[im]; (void)rI;
#else
                buf_ld2(rI, A.vo[im], 0, g0x, g0y);
                buf_ld2(rI, A.vo[im], ipb, g1x, g1y);
#endif
                G[im][0].x = g0x; G[im][0].y = g0y; G[im][1].x = g1x; G[im][1].y = g1y;
            }
        };
        auto wblend = [&](auto pc, const WAddr &A, const lk_d2 (&G)[2][2]) __attribute__((always_inline)) {     // blend the gathered taps of a step -> LDS plane
            constexpr int WP = decltype(pc)::value;
#pragma unroll
            for (int im = 0; im < 2; ++im) {
                const char *wt = reinterpret_cast<const char *>(&s_wt[0][0]) + A.wi[im];
                const lk_d2 w01 = *reinterpret_cast<const lk_d2 *>(wt), w23 = *reinterpret_cast<const lk_d2 *>(wt + 16);
                lk_d2 top = G[im][0], bot = G[im][1];
                if (__builtin_amdgcn_ballot_w64((A.dx[im] | A.dy[im]) != 0) != 0) {   // some lane's taps leave the image (wave-uniform, no memory instruction)
                    const int dx = A.dx[im], dy = A.dy[im];
                    const lk_d2 z = {0.0, 0.0};
                    const lk_d2 t0 = top, t1 = bot;
                    top = dy == 0 ? t0 : (dy == 1 ? t1 : z);
                    bot = dy == 0 ? t1 : (dy == -1 ? t0 : z);
                    const lk_d2 a0 = top, a1 = bot;
                    top.x = dx == 0 ? a0.x : (dx == 1 ? a0.y : 0.0); top.y = dx == 0 ? a0.y : (dx == -1 ? a0.x : 0.0);
                    bot.x = dx == 0 ? a1.x : (dx == 1 ? a1.y : 0.0); bot.y = dx == 0 ? a1.y : (dx == -1 ? a1.x : 0.0);
                }
                s_w[WP][E][im][g] = top.x * w01.x + top.y * w01.y + bot.x * w23.x + bot.y * w23.y;
            }
        };
        WAddr cur, nxt;
        {   // prologue: the carried rows and the rotations of W-steps 0 and 1 straight from memory (from step 2 on: F's service)
            const UpSel1 us = make_upsel1(xcol, Wc);
            const int cpub = (int)a.cu.pitch * 8, cpvb = (int)a.cv.pitch * 8;
            const __amdgpu_buffer_rsrc_t rCU = buf_rsrc(a.cu.at(pair) + (long long)c_lo * a.cu.pitch, ((long long)(Hc - 1 - c_lo) * a.cu.pitch + Wc) * 8);
            const __amdgpu_buffer_rsrc_t rCV = buf_rsrc(a.cv.at(pair) + (long long)c_lo * a.cv.pitch, ((long long)(Hc - 1 - c_lo) * a.cv.pitch + Wc) * 8);
            auto srow = [&](int sr, double &su, double &sv) __attribute__((always_inline)) {
                su = pyrup_h1v(buf_ld(rCU, us.jm, sr * cpub), buf_ld(rCU, us.jc, sr * cpub), buf_ld(rCU, us.jp, sr * cpub), us);
                sv = pyrup_h1v(buf_ld(rCV, us.jm, sr * cpvb), buf_ld(rCV, us.jc, sr * cpvb), buf_ld(rCV, us.jp, sr * cpvb), us);
            };
#pragma unroll
            for (int q = 0; q < 3; ++q) srow(up_srow_w(ib - 1 + q) - c_lo, hu[q], hv[q]);
            double nu, nv;
            srow(rot_row_off(0), nu, nv); waddr(0, nu, nv, cur);
            wissue(cur, Ga);
            srow(rot_row_off(1), nu, nv); waddr(1, nu, nv, nxt);
        }
        __syncthreads();                             // F's service has published the row W-step 2 rotates in
        // step t: start the gathers of step t + 1, make the addresses of step t + 2, and only then blend step t's taps (gathered
        // since the start of step t - 1) into the LDS plane F reads next step
        auto wstep = [&](auto pc, int t, lk_d2 (&Gc)[2][2], lk_d2 (&Gn)[2][2]) __attribute__((always_inline)) {
            constexpr int WP = decltype(pc)::value;
            wissue(nxt, Gn);
            const double nu = s_c[WP ^ 1][0][g], nv = s_c[WP ^ 1][1][g];   // written by F in step t - 1: the row step t + 2 rotates in
            WAddr nw;
            waddr(t + 2, nu, nv, nw);
            wblend(pc, cur, Gc);
            cur = nxt; nxt = nw;
            __syncthreads();
        };
        for (int t = 0; t < NS; t += 2) {                               // NS is even
            wstep(std::integral_constant<int, 0>{}, t, Ga, Gb);
            wstep(std::integral_constant<int, 1>{}, t + 1, Gb, Ga);
        }
        };
        if (widx < 2) wrole(std::integral_constant<int, 0>{});          // (wave-uniform)
        else wrole(std::integral_constant<int, 1>{});
      }
    } else if (wv >= C::NB + 5) {
        // =========================================================================================
        // F waves: two adjacent columns per lane -> products
        // =========================================================================================
        // the one wave that feeds the block's pipeline from memory issues ahead of the seven that consume (A/B: level-0
        // launch 0.80 -> 0.77 ms; raising B instead, or both, gains nothing)
        __builtin_amdgcn_s_setprio(LK_SPLIT_PRIO_F);
        const int m = ltid & 63;                 // lane: block columns 2(m-1), 2(m-1)+1; lane 0 = the pair left of the block
        [[maybe_unused]] const int fe = wv - (C::NB + 5);   // two F waves: this one makes the products of row fe of every step
      if constexpr (FW) {
        // fused warp: the rows come from the W waves through LDS (plane (t - 1) & 1 in pipeline step t); F-step s = t - 2.
        // Steps 0 and 1 run the same code on rows nobody consumes (step 1 leaves row yy0 - 1 behind as the previous row).
        const int xa = x0 + 2 * (m - 1);
        const bool col_in = xa >= 0 && xa < W;
        const bool all_col_in = __builtin_amdgcn_ballot_w64(!col_in) == 0;
        const int p_chunk = PADJ + m - 1;
        auto left_tap = [&](double vb) __attribute__((always_inline)) { return dpp_wave_shr1(vb); };
        // coarse-row service for the W waves: the horizontally filtered coarse row (cv::pyrUp's horizontal pass, both flow
        // planes, both columns of this lane's pair) that W-step t + 3 rotates in, published in step t; its taps are fetched
        // one step ahead.  (F has no other memory traffic in this mode, and the W lanes of a pair share their taps.)
        const int Hc = H >> 1, Wc = W >> 1;
        const int xc = min(max(xa, 0), W - 2);
        const UpSel1 usA = make_upsel1(xc, Wc), usB = make_upsel1(xc + 1, Wc);
        const int cpub = (int)a.cu.pitch * 8, cpvb = (int)a.cv.pitch * 8;
        const __amdgpu_buffer_rsrc_t rCU = buf_rsrc(a.cu.at(pair) + (long long)c_lo * a.cu.pitch, ((long long)(Hc - 1 - c_lo) * a.cu.pitch + Wc) * 8);
        const __amdgpu_buffer_rsrc_t rCV = buf_rsrc(a.cv.at(pair) + (long long)c_lo * a.cv.pitch, ((long long)(Hc - 1 - c_lo) * a.cv.pitch + Wc) * 8);
        double tpf[6];
        // (two F waves: wave 0 serves the u plane, wave 1 the v plane)
        auto svc_load = [&](int t) __attribute__((always_inline)) {     // taps of the row W-step t rotates in
            const int sr = rot_row_off(t), su = sr * cpub, sv = sr * cpvb;
            if (C::NF == 1 || fe == 0) { tpf[0] = buf_ld(rCU, usA.jm, su); tpf[1] = buf_ld(rCU, usA.jc, su); tpf[2] = buf_ld(rCU, usA.jp, su); }
            if (C::NF == 1 || fe == 1) { tpf[3] = buf_ld(rCV, usA.jm, sv); tpf[4] = buf_ld(rCV, usA.jc, sv); tpf[5] = buf_ld(rCV, usA.jp, sv); }
        };
        auto svc_publish = [&](auto pc) __attribute__((always_inline)) {
            constexpr int P = decltype(pc)::value;
            if (C::NF == 1 || fe == 0) {
                lk_d2 ou;
                ou.x = pyrup_h1v(tpf[0], tpf[1], tpf[2], usA); ou.y = pyrup_h1v(tpf[0], tpf[1], tpf[2], usB);
                *reinterpret_cast<lk_d2 *>(&s_c[P][0][2 * m]) = ou;
            }
            if (C::NF == 1 || fe == 1) {
                lk_d2 ov;
                ov.x = pyrup_h1v(tpf[3], tpf[4], tpf[5], usA); ov.y = pyrup_h1v(tpf[3], tpf[4], tpf[5], usB);
                *reinterpret_cast<lk_d2 *>(&s_c[P][1][2 * m]) = ov;
            }
        };
        svc_load(2); svc_publish(std::integral_constant<int, 1>{}); svc_load(3);
        __syncthreads();
        double pl[2] = {0, 0}, pa[2] = {0, 0}, pb[2] = {0, 0};
        auto fstepw = [&](auto uc, int t) __attribute__((always_inline)) {
            constexpr int SP = decltype(uc)::value & 1, WP = SP ^ 1;
            lk_d2 rw[2][2];
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int im = 0; im < 2; ++im) rw[e][im] = *reinterpret_cast<const lk_d2 *>(&s_w[WP][e][im][2 * m]);
            [&]<int... Es>(std::integer_sequence<int, Es...>) __attribute__((always_inline)) {
                ([&] {
                    constexpr int E = Es;
                    double cl[2], ca[2], cb[2];
#pragma unroll
                    for (int im = 0; im < 2; ++im) { ca[im] = rw[E][im].x; cb[im] = rw[E][im].y; cl[im] = left_tap(cb[im]); }
                    const int yy = yy0 + 2 * (t - 2) + E;
                    const bool row_in = yy >= 0 && yy < H;
                    if (C::NF == 1 || fe == E) {                // (wave-uniform: two F waves make one row of the pair each)
#ifdef LK_ABL_F   /* timing-only ablation: results wrong */
                    double fxa = pl[0] + ca[0], fya = pa[0] + cl[1], fxb = pb[0], fyb = cb[1];
                    const double fta = ca[1], ftb = cb[0];
#else
                    double fxa = f2d_x(pl[0], pa[0], cl[0], ca[0]) + f2d_x(pl[1], pa[1], cl[1], ca[1]);
                    double fya = f2d_y(pl[0], pa[0], cl[0], ca[0]) + f2d_y(pl[1], pa[1], cl[1], ca[1]);
                    const double fta = f2d_neg(pl[0], pa[0], cl[0], ca[0]) + f2d_pos(pl[1], pa[1], cl[1], ca[1]);
                    double fxb = f2d_x(pa[0], pb[0], ca[0], cb[0]) + f2d_x(pa[1], pb[1], ca[1], cb[1]);
                    double fyb = f2d_y(pa[0], pb[0], ca[0], cb[0]) + f2d_y(pa[1], pb[1], ca[1], cb[1]);
                    const double ftb = f2d_neg(pa[0], pb[0], ca[0], cb[0]) + f2d_pos(pa[1], pb[1], ca[1], cb[1]);
#endif
                    if (!(row_in && all_col_in)) {              // rows / columns outside the image contribute 0 (wave-uniform test)
                        const bool keep = row_in && col_in;
                        fxa = keep ? fxa : 0.0; fya = keep ? fya : 0.0;
                        fxb = keep ? fxb : 0.0; fyb = keep ? fyb : 0.0;
                    }
                    auto put = [&](auto fc, double xa_, double xb_) __attribute__((always_inline)) {
                        lk_d2 o; o.x = xa_; o.y = xb_;
                        *reinterpret_cast<lk_d2 *>(&s_prod[SP][E][decltype(fc)::value][2 * p_chunk]) = o;
                    };
                    put(std::integral_constant<int, 0>{}, fxa * fxa, fxb * fxb);
                    put(std::integral_constant<int, 1>{}, fya * fya, fyb * fyb);
                    put(std::integral_constant<int, 2>{}, fxa * fya, fxb * fyb);
                    put(std::integral_constant<int, 3>{}, fxa * fta, fxb * ftb);
                    put(std::integral_constant<int, 4>{}, fya * fta, fyb * ftb);
                    }
#pragma unroll
                    for (int im = 0; im < 2; ++im) { pl[im] = cl[im]; pa[im] = ca[im]; pb[im] = cb[im]; }
                }(), ...);
            }(std::make_integer_sequence<int, 2>{});
            svc_publish(std::integral_constant<int, SP>{});             // the row W-step t + 3 rotates in (W reads it in step t + 1)
            svc_load(t + 4);
            __syncthreads();
        };
        for (int t = 0; t < NS; t += 2) {
            fstepw(std::integral_constant<int, 0>{}, t);
            fstepw(std::integral_constant<int, 1>{}, t + 1);
        }
      } else {
        const int xa = x0 + 2 * (m - 1);             // even.  W even: the pair is all in or all out; W odd (a coarsest level): the pair
                                                     // (W - 1, W) is half in -- its load is clamped to (W - 2, W - 1), see half_fix below
        const int pitchb = (int)a.i1.pitch * 8;
        const bool col_in = xa >= 0 && xa < W, col_in_b = xa >= 0 && xa + 1 < W;
        const bool all_col_in = __builtin_amdgcn_ballot_w64(!col_in_b) == 0;
        const int xc = min(max(xa, 0), W - 2);
        // W odd: the lane of the half pair has loaded (pixel W - 2, pixel W - 1) and wants pixel W - 1 as its FIRST column (its second
        // column, W, lies outside: zero products).  Wave-uniform: one wave of the last column block has such a lane, no wave when W is even.
        const bool half_lane = xa == W - 1;
        const bool half_wave = __builtin_amdgcn_ballot_w64(half_lane) != 0;
        auto half_fix = [&](double &va, double vb) __attribute__((always_inline)) { if (half_wave) va = half_lane ? vb : va; };
        const int row_lo = max(y0 - R - PF, 0);
        const long long img_bytes = ((long long)(H - 1 - row_lo) * a.i1.pitch + W) * 8;
        const __amdgpu_buffer_rsrc_t rI1 = buf_rsrc(a.i1.at(pair) + (long long)row_lo * a.i1.pitch, img_bytes);
        const __amdgpu_buffer_rsrc_t rI2 = buf_rsrc(a.i2.at(pair) + (long long)row_lo * a.i2.pitch, img_bytes);
        const int vxc = xc * 8;
        const int p_chunk = PADJ + m - 1;            // (lane 0 lands in the last left pad chunk: real products, never summed into an output)

        // byte offset of image row yy: one reflection (exact for every row an output needs: H > R + PF + 1),
        // then clamped (prefetches of the idle tail steps run past the last needed row)
        auto img_row_off = [&](int yy) {
            int rr = yy < 0 ? -yy : yy;
            rr = min(rr, 2 * H - 2 - rr);
            rr = max(rr, row_lo);
            return (rr - row_lo) * pitchb;
        };
        // left derivative tap of column xa for a row whose pair is (.., vb) in this lane: the left lane's second pixel.
        // At image column 0 the left lane's pair is clamped to columns (0, 1), so it hands over column 1 = the
        // reflect-101 tap (lane 0 of a wave never produces an output column: it is the halo pair).
        auto left_tap = [&](double vb) __attribute__((always_inline)) { return dpp_wave_shr1(vb); };
        // previous row (p*) and current row (c*): left tap l, columns a and b; [0] image 1, [1] image 2
        double pl[2], pa[2], pb[2], cl[2], ca[2], cb[2];
        {
            int ro = img_row_off(yy0 - 1);
            buf_ld2(rI1, vxc, ro, pa[0], pb[0]); buf_ld2(rI2, vxc, ro, pa[1], pb[1]);
            ro = img_row_off(yy0);
            buf_ld2(rI1, vxc, ro, ca[0], cb[0]); buf_ld2(rI2, vxc, ro, ca[1], cb[1]);
            half_fix(pa[0], pb[0]); half_fix(pa[1], pb[1]); half_fix(ca[0], cb[0]); half_fix(ca[1], cb[1]);
            pl[0] = left_tap(pb[0]); pl[1] = left_tap(pb[1]);
            cl[0] = left_tap(cb[0]); cl[1] = left_tap(cb[1]);
        }
        double qa[PF][2], qb[PF][2];                        // queue slot of marched row i = i % PF
#pragma unroll
        for (int r = 1; r < PF; ++r) {
            const int o = img_row_off(yy0 + r);
            buf_ld2(rI1, vxc, o, qa[r][0], qb[r][0]);
            buf_ld2(rI2, vxc, o, qa[r][1], qb[r][1]);
        }
        qa[0][0] = qa[0][1] = qb[0][0] = qb[0][1] = 0.0;

        // one marched row i (queue slot QS = i % PF, static): prefetch row i + PF into the slot the last rotate
        // freed, products of row i -> LDS plane (SP, E), rotate row i + 1 in
        auto frow = [&](auto uc, auto ec, int i) __attribute__((always_inline)) {
            constexpr int U = decltype(uc)::value, E = decltype(ec)::value;
            constexpr int SP = U & 1;                        // LDS plane of the step
            constexpr int QS = (2 * U + E) % PF;
            {
                const int rn = img_row_off(yy0 + i + PF);
                buf_ld2(rI1, vxc, rn, qa[QS][0], qb[QS][0]);
                buf_ld2(rI2, vxc, rn, qa[QS][1], qb[QS][1]);
            }
            const int yy = yy0 + i;
            const bool row_in = yy >= 0 && yy < H;
            if (C::NF == 1 || fe == E) {                     // (wave-uniform; no memory instruction inside)
            double fxa = f2d_x(pl[0], pa[0], cl[0], ca[0]) + f2d_x(pl[1], pa[1], cl[1], ca[1]);
            double fya = f2d_y(pl[0], pa[0], cl[0], ca[0]) + f2d_y(pl[1], pa[1], cl[1], ca[1]);
            const double fta = f2d_neg(pl[0], pa[0], cl[0], ca[0]) + f2d_pos(pl[1], pa[1], cl[1], ca[1]);
            double fxb = f2d_x(pa[0], pb[0], ca[0], cb[0]) + f2d_x(pa[1], pb[1], ca[1], cb[1]);
            double fyb = f2d_y(pa[0], pb[0], ca[0], cb[0]) + f2d_y(pa[1], pb[1], ca[1], cb[1]);
            const double ftb = f2d_neg(pa[0], pb[0], ca[0], cb[0]) + f2d_pos(pa[1], pb[1], ca[1], cb[1]);
            if (!(row_in && all_col_in)) {                  // rows / columns outside the image contribute 0 (wave-uniform test)
                const bool keep = row_in && col_in, keep_b = row_in && col_in_b;
                fxa = keep ? fxa : 0.0; fya = keep ? fya : 0.0;
                fxb = keep_b ? fxb : 0.0; fyb = keep_b ? fyb : 0.0;
            }
            lk_d2 o;
            // per field: the two products as one 16-byte store, their sum (what S adds up for the shared part of the
            // two windows of a column pair) as one 8-byte store
            auto put = [&](auto fc, double xa_, double xb_) __attribute__((always_inline)) {
                constexpr int FI = decltype(fc)::value;
                o.x = xa_; o.y = xb_;
                *reinterpret_cast<lk_d2 *>(&s_prod[SP][E][FI][2 * p_chunk]) = o;
                if constexpr (C::QSUM) s_q[SP][E][FI][p_chunk] = xa_ + xb_;
            };
            put(std::integral_constant<int, 0>{}, fxa * fxa, fxb * fxb);
            put(std::integral_constant<int, 1>{}, fya * fya, fyb * fyb);
            put(std::integral_constant<int, 2>{}, fxa * fya, fxb * fyb);
            put(std::integral_constant<int, 3>{}, fxa * fta, fxb * ftb);
            put(std::integral_constant<int, 4>{}, fya * fta, fyb * ftb);
            }
            // rotate: the only wait for the prefetch
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                pl[g] = cl[g]; pa[g] = ca[g]; pb[g] = cb[g];
                ca[g] = qa[(QS + 1) % PF][g]; cb[g] = qb[(QS + 1) % PF][g];
                half_fix(ca[g], cb[g]);
                cl[g] = left_tap(cb[g]);
            }
        };
        auto fstep = [&](auto spc, int st) __attribute__((always_inline)) {
            frow(spc, std::integral_constant<int, 0>{}, 2 * st);
            frow(spc, std::integral_constant<int, 1>{}, 2 * st + 1);
            __syncthreads();
        };
        for (int st = 0; st < NS; st += UN) {                // NS is a multiple of UN: no guards inside a group
            [&]<int... Us>(std::integer_sequence<int, Us...>) __attribute__((always_inline)) {
                (fstep(std::integral_constant<int, Us>{}, st + Us), ...);
            }(std::make_integer_sequence<int, UN>{});
        }
      }
    } else if (wv < C::NB) {
        // =========================================================================================
        // B waves: one column per lane: window sums -> flow
        // =========================================================================================
#ifdef LK_SPLIT_PRIO_B
        __builtin_amdgcn_s_setprio(LK_SPLIT_PRIO_B);
#endif
        const int t = ltid;                          // block column 0 .. TXC-1
        const int x = x0 + t;
        const int xc = min(max(x, 0), W - 1);
        // output column of this block that is NOT in the frame of width R just inside the image: the frame (clipped
        // windows and the outermost ring) belongs to lk_border_kernel, which runs concurrently on a side stream
        const bool is_out = t >= HL && t < HL + XO && x >= R && x < W - R;
        const int s_slot = KH == 1 ? t : 2 * (((t % K) >> 1) * C::SSTR + t / K) + (t & 1);

        [[maybe_unused]] UpSel1 us{};
        [[maybe_unused]] double hu[3] = {0, 0, 0}, hv[3] = {0, 0, 0};   // carried coarse rows (rotated every even row)
        // taps of the next coarse rows (u: 0..2, v: 3..5): a queue TD row pairs deep.  A step is shorter than an L2 / HBM
        // round trip, and vmcnt is in-order, so a tap that was fetched one step ahead also waited for every older
        // store of this wave (timing-only ablations: removing the stores alone saved 0.12 of 0.69 ms)
        constexpr int TD = LK_SPLIT_TD;
        [[maybe_unused]] double tp[TD][6];
#pragma unroll
        for (int q = 0; q < TD; ++q)
#pragma unroll
            for (int k = 0; k < 6; ++k) tp[q][k] = 0.0;
        [[maybe_unused]] const int Hc = H >> 1;
        [[maybe_unused]] const int cpub = (int)a.cu.pitch * 8, cpvb = (int)a.cv.pitch * 8;
        [[maybe_unused]] const int crow_lo = max((y0 >> 1) - 1, 0);
        auto up_srow = [&](int k) { return k < 0 ? min(1, Hc - 1) : (k > Hc - 1 ? Hc - 1 : k); };
        const int upb = (int)a.u.pitch * 8, vpb = (int)a.v.pitch * 8;
        const __amdgpu_buffer_rsrc_t rU = buf_rsrc(a.u.at(pair) + (long long)y0 * a.u.pitch, ((long long)(nrows - 1) * a.u.pitch + W) * 8);
        const __amdgpu_buffer_rsrc_t rV = buf_rsrc(a.v.at(pair) + (long long)y0 * a.v.pitch, ((long long)(nrows - 1) * a.v.pitch + W) * 8);
        // (without HAS_UP the coarse descriptors are never used; they alias the outputs so that they are well-formed)
        [[maybe_unused]] const __amdgpu_buffer_rsrc_t rCU =
            HAS_UP ? buf_rsrc(a.cu.at(pair) + (long long)crow_lo * a.cu.pitch, ((long long)(Hc - 1 - crow_lo) * a.cu.pitch + (W >> 1)) * 8) : rU;
        [[maybe_unused]] const __amdgpu_buffer_rsrc_t rCV =
            HAS_UP ? buf_rsrc(a.cv.at(pair) + (long long)crow_lo * a.cv.pitch, ((long long)(Hc - 1 - crow_lo) * a.cv.pitch + (W >> 1)) * 8) : rV;
        // byte offsets (u plane, v plane) of the coarse row whose taps output row y + 2 (y odd) consumes: s(((y + 2) >> 1) + 2)
        auto tap_off = [&](int y, int &su, int &sv) {
            const int sr = max(min(((y + 2) >> 1) + 2, Hc - 1) - crow_lo, 0);
            su = sr * cpub; sv = sr * cpvb;
        };
        if (HAS_UP) {
            us = make_upsel1(xc, W >> 1);
            const int i0 = y0 >> 1;
            // slots in the order the unconditional rotate of the first (even) row expects: (s(i0+1), s(i0-1), s(i0))
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int sr = up_srow(i0 - 1 + s) - crow_lo;
                hu[(s + 1) % 3] = pyrup_h1v(buf_ld(rCU, us.jm, sr * cpub), buf_ld(rCU, us.jc, sr * cpub), buf_ld(rCU, us.jp, sr * cpub), us);
                hv[(s + 1) % 3] = pyrup_h1v(buf_ld(rCV, us.jm, sr * cpvb), buf_ld(rCV, us.jc, sr * cpvb), buf_ld(rCV, us.jp, sr * cpvb), us);
            }
#pragma unroll
            for (int q = 0; q < TD - 1; ++q) {                  // taps that the odd rows of pairs 0 .. TD-2 consume -> slots 0 .. TD-2
                int su, sv;
                tap_off(y0 + 2 * q - 1, su, sv);
                tp[q][0] = buf_ld(rCU, us.jm, su); tp[q][1] = buf_ld(rCU, us.jc, su); tp[q][2] = buf_ld(rCU, us.jp, su);
                tp[q][3] = buf_ld(rCV, us.jm, sv); tp[q][4] = buf_ld(rCV, us.jc, sv); tp[q][5] = buf_ld(rCV, us.jp, sv);
            }
        }
        const int vxo = x * 8;
        int sou = 0, sov = 0;                                   // byte offsets of output row r in the u / v strips

        // One step = the output row pair (y, y + 1), y = y0 + 2p even, at static pair position P2 = p mod TD (tap
        // slot; LDS step parity = (p + R) & 1).  Straight-line: every vector-memory instruction is issued
        // unconditionally (see the file header); the two solves are independent and interleave.
#ifdef LK_ABL_B_NOSTORE   /* timing-only ablation: nothing is stored */
        const int vo_lane = (int)0x80000000; (void)vxo;
#else
        const int vo_lane = is_out ? vxo : (int)0x80000000;   // out of range: the store is dropped
#endif
        auto solve5 = [&](const double (&S)[5], double &u, double &v) __attribute__((always_inline)) {
            const double det = S[0] * S[1] - S[2] * S[2];
            const double un = S[2] * S[4] - S[1] * S[3];
            const double vn = S[3] * S[2] - S[0] * S[4];
            const bool nz = det != 0.0;                     // cv::divide: x/0 -> 0
            double qu, qv;
#ifdef LK_ABL_B   /* timing-only ablation: results wrong (kept finite and tiny so that the next level's gathers stay local) */
            qu = un * 1e-300; qv = vn * 1e-300; (void)det;
#else
            lk_div2(un, vn, det, qu, qv);                   // both quotients correctly rounded, one reciprocal iteration (lk_device.h)
#endif
            u = nz ? qu : 0.0;
            v = nz ? qv : 0.0;
        };
        // (pixels whose window is clipped by the image border get m(y,x) * WIN^2 instead of a window sum,
        // SerialPyr.cpp:94-95, and the outermost ring gets 0: lk_border_kernel writes that whole frame, this kernel
        // stores none of it -- a load under a condition in this loop would make hipcc's vmcnt counting conservative
        // and drain the wave's stores at every step)
        auto solve_row = [&](int, const double (&S)[5], double &u, double &v) __attribute__((always_inline)) {
            solve5(S, u, v);
        };
        auto bpair = [&](auto pc2, int p) __attribute__((always_inline)) {
            constexpr int P2 = decltype(pc2)::value;        // p mod TD
            constexpr int SPB = (P2 + R) & 1;
            constexpr int TN = (P2 + TD - 1) % TD;          // the slot this pair's own taps just left
            const int y = y0 + 2 * p;
#ifdef LK_ABL_B_NOLOAD   /* timing-only ablation: the coarse taps are not fetched (results wrong) */
            if constexpr (false) {
#else
            if constexpr (HAS_UP) {                         // taps consumed by the odd row of pair p + TD - 1
#endif
                int su, sv;
                tap_off(y + 2 * (TD - 1) - 1, su, sv);
                tp[TN][0] = buf_ld(rCU, us.jm, su); tp[TN][1] = buf_ld(rCU, us.jc, su); tp[TN][2] = buf_ld(rCU, us.jp, su);
                tp[TN][3] = buf_ld(rCV, us.jm, sv); tp[TN][4] = buf_ld(rCV, us.jc, sv); tp[TN][5] = buf_ld(rCV, us.jp, sv);
            }
            double S0[5], S1[5];
#pragma unroll
            for (int f = 0; f < 5; ++f) { S0[f] = s_sums[SPB][0][f][s_slot]; S1[f] = s_sums[SPB][1][f][s_slot]; }
            double ue, ve, uo, vo;
            solve_row(y, S0, ue, ve);
            solve_row(y + 1, S1, uo, vo);
            if constexpr (HAS_UP) {
                {                                           // even row: rotate in the coarse row turned into slot 0 last pair
                    const double nu = hu[0], nv = hv[0];
                    hu[0] = hu[1]; hu[1] = hu[2]; hu[2] = nu;
                    hv[0] = hv[1]; hv[1] = hv[2]; hv[2] = nv;
                    ue += (hu[0] + hu[1] * 6 + hu[2]) * (1.0 / 64);
                    ve += (hv[0] + hv[1] * 6 + hv[2]) * (1.0 / 64);
                }
                uo += ((hu[1] + hu[2]) * 4) * (1.0 / 64);
                vo += ((hv[1] + hv[2]) * 4) * (1.0 / 64);
                hu[0] = pyrup_h1v(tp[P2][0], tp[P2][1], tp[P2][2], us);   // next coarse row, used from the next pair on
                hv[0] = pyrup_h1v(tp[P2][3], tp[P2][4], tp[P2][5], us);
            }
            // rows of the frame (wave-uniform test): out-of-range offset, the store is dropped
            const int voe = (y >= R && y < H - R) ? vo_lane : (int)0x80000000;
            const int voo = (y + 1 >= R && y + 1 < H - R) ? vo_lane : (int)0x80000000;
            buf_st(rU, voe, sou, ue);
            buf_st(rV, voe, sov, ve);
            buf_st(rU, voo, sou + upb, uo);
            buf_st(rV, voo, sov + vpb, vo);
            sou += 2 * upb; sov += 2 * vpb;
        };
        auto bstep = [&](auto pc2, int p) __attribute__((always_inline)) {
            bpair(pc2, p);
            __syncthreads();
        };
        for (int k = 0; k < R + 2 + C::LAG + (FW ? 1 : 0); ++k) __syncthreads();   // (FW: W / F prologue,) pipeline fill: no window sums yet
        const int npairs = (nrows + 1) / 2;
        int p = 0;
        for (; p + TD <= npairs; p += TD) {                     // whole groups
            [&]<int... Ps>(std::integer_sequence<int, Ps...>) __attribute__((always_inline)) {
                (bstep(std::integral_constant<int, Ps>{}, p + Ps), ...);
            }(std::make_integer_sequence<int, TD>{});
        }
        [&]<int... Ps>(std::integer_sequence<int, Ps...>) __attribute__((always_inline)) {   // last, partial group
            ((p + Ps < npairs ? (bstep(std::integral_constant<int, Ps>{}, p + Ps), 0) : 0), ...);
        }(std::make_integer_sequence<int, TD - 1>{});
        for (int k = ns; k < NS; ++k) __syncthreads();          // idle tail steps
    } else {
        // =========================================================================================
        // S waves: one field each, K adjacent columns per lane
        // =========================================================================================
        const int f = wv - C::NB;
        const int j = threadIdx.x & 63;
        // Horizontal windows of the lane's column pair (2j, 2j+1) from the pair sums q and two single products:
        //   R odd : shared = q[j-(R-1)/2 .. j+(R-1)/2];          out0 = b[j-(R+1)/2] + shared;  out1 = shared + a[j+(R+1)/2]
        //   R even: shared = q[j-(R/2-1) .. j+(R/2-1)];  out0 = q[j-R/2] + shared + a[j+R/2];  out1 = b[j-R/2] + shared + q[j+R/2]
        // (R or R+1 eight-byte LDS reads + 2 and R+1 or R+2 additions per row, against 2R+2 values and 2R+1 additions
        // from the raw products; additions only).  NQ values of q starting at lane offset QLO, singles at offsets -HS, +HS.
        constexpr bool RODD = (R & 1) != 0;
        constexpr int HS = RODD ? (R + 1) / 2 : R / 2;       // lane offset of the two single products
        constexpr int QH = RODD ? (R - 1) / 2 : R / 2;       // q values are read for lane offsets -QH .. +QH
        constexpr int NQ = 2 * QH + 1;
        static_assert(K == 2 && HS <= PADJ, "pad lanes cover the taps");
        struct SRow { double q[NQ], sb, sa; };
        constexpr bool QDPP = LK_SPLIT_QDPP && !(FW && LK_SPLIT_FW_NOQDPP && WIN <= 21);   // (fused warp, A/B: pair sums by LDS reads again)
        constexpr bool STREAMED = QDPP && !C::QSUM && WIN > 21;   // (see sload)
        double ring[WIN][K], pre[K];
#pragma unroll
        for (int q = 0; q < WIN; ++q)
#pragma unroll
            for (int c = 0; c < K; ++c) ring[q][c] = 0.0;
#pragma unroll
        for (int c = 0; c < K; ++c) pre[c] = 0.0;

        auto sload = [&](auto spc, auto ec, SRow &r) __attribute__((always_inline)) {
            constexpr int SPP = decltype(spc)::value, E = decltype(ec)::value;   // LDS plane (step parity), row of the step
            const double *pp = &s_prod[SPP][E][f][2 * (PADJ + j)];
            if constexpr (!C::QSUM && QDPP) {
                // The lane reads ONLY its own pair and forms its pair sum; the neighbours' pair sums -- the same values, each
                // formed once by the lane that owns the pair -- arrive by DPP wave shifts, one lane per shift.  Against NQ
                // sixteen-byte LDS reads and NQ additions per row: 1 read, 1 addition and 2 (NQ - 1) 32-bit moves; the LDS pipe
                // is this kernel's busiest unit.  Same values in the same tree: bit-identical.  Lanes within
                // QH of a wave edge receive zeros -- they are halo lanes, never output lanes (HL / 2 >= QH).
                const lk_d2 c2 = *reinterpret_cast<const lk_d2 *>(pp);
                const double q0 = c2.x + c2.y;
                double l = q0, g = q0;
                if constexpr (STREAMED) {
                    // windows above 21 x 21: the shared part of the two windows is accumulated as the pair sums arrive (centre
                    // outwards, left before right) instead of being kept for a tree: 3 live values instead of NQ -- the 2 NQ
                    // registers this frees are what lets 23 x 23 run two workgroups per CU (ring 88 + the rest <= 128 VGPRs)
                    double acc = q0;
#pragma unroll
                    for (int d = 1; d <= QH; ++d) {
                        l = dpp_from_left(l); g = dpp_from_right(g);
                        if (RODD || d < QH) acc = (acc + l) + g;
                        else { r.q[0] = l; r.q[NQ - 1] = g; }        // R even: the outermost pair sums belong to one window each
                    }
                    r.q[QH] = acc;
                } else {
                    r.q[QH] = q0;
#pragma unroll
                    for (int d = 1; d <= QH; ++d) {
                        l = dpp_from_left(l); r.q[QH - d] = l;       // q of pair j - d
                        g = dpp_from_right(g); r.q[QH + d] = g;      // q of pair j + d
                    }
                }
                r.sb = pp[-2 * HS + 1];
                r.sa = pp[2 * HS];
                return;
            }
            if constexpr (!C::QSUM) {                        // the pairs themselves (16-byte reads), summed here: F, the one wave
#pragma unroll                                               // that feeds the block, has the longest instruction stream
                for (int d = 0; d < NQ; ++d) {
                    const lk_d2 c2 = *reinterpret_cast<const lk_d2 *>(pp + 2 * (d - QH));
                    r.q[d] = c2.x + c2.y;
                }
                r.sb = pp[-2 * HS + 1];
                r.sa = pp[2 * HS];
                return;
            }
            const double *qq = &s_q[C::QSUM ? SPP : 0][C::QSUM ? E : 0][C::QSUM ? f : 0][C::QSUM ? PADJ + j : 0];
#pragma unroll
            for (int d = 0; d < NQ; ++d) r.q[d] = qq[d - QH];
            r.sb = pp[-2 * HS + 1];                          // b of pair j - HS
            r.sa = pp[2 * HS];                               // a of pair j + HS
        };
        // horizontal windows + vertical van Herk step at static chunk position KC -> the K window sums of row i
        auto ssum = [&](auto kc, const SRow &r, double (&S)[K]) __attribute__((always_inline)) {
            constexpr int KC = decltype(kc)::value;
            double hs[K];
            // the part the two windows share: up to 21 x 21 a tree over the pair sums; above, centre outwards, left before right --
            // the order in which the STREAMED form meets them -- in EVERY form of the kernel (published pair sums at 29 x 29 and
            // 31 x 31, fused warp, LK_SPLIT_QDPP = 0), so that a window's bits do not depend on which form a level's size selects
            double shared;
            if constexpr (STREAMED) shared = r.q[QH];
            else if constexpr (WIN > 21) {
                shared = r.q[QH];
#pragma unroll
                for (int d = 1; d <= QH; ++d)
                    if (RODD || d < QH) shared = (shared + r.q[QH - d]) + r.q[QH + d];
            } else shared = RODD ? split_tree<0, NQ - 1>(r.q) : split_tree<1, NQ - 2>(r.q);
            if constexpr (RODD) {
                hs[0] = r.sb + shared;
                hs[1] = shared + r.sa;
            } else {
                hs[0] = (r.q[0] + shared) + r.sa;
                hs[1] = (r.sb + shared) + r.q[NQ - 1];
            }
#pragma unroll
            for (int c = 0; c < K; ++c) {
                pre[c] = (KC == 0) ? hs[c] : pre[c] + hs[c];
                S[c] = (KC < WIN - 1) ? ring[(KC + 1) % WIN][c] + pre[c] : pre[c];
                if (KC > 0) ring[KC][c] = hs[c];
            }
            if (KC == WIN - 1) {
#pragma unroll
                for (int q = WIN - 2; q >= 1; --q)
#pragma unroll
                    for (int c = 0; c < K; ++c) ring[q][c] += ring[q + 1][c];
            }
        };
        // (rows before the first full vertical window are stored too: B reads a sums plane only from marched row 2R on,
        // by which time every plane has been overwritten with real sums -- an unconditional store costs no test)
        auto sstore = [&](auto spc, auto ec, const double (&S)[K]) __attribute__((always_inline)) {
            constexpr int SPP = decltype(spc)::value, E = decltype(ec)::value;
            {
                double *sp = &s_sums[SPP][E][f][0];
#pragma unroll
                for (int h = 0; h < KH; ++h) {
                    lk_d2 o; o.x = S[2 * h]; o.y = S[2 * h + 1];
                    *reinterpret_cast<lk_d2 *>(sp + 2 * (h * C::SSTR + j)) = o;
                }
            }
        };
        // one step = two marched rows at static position T of the unrolled body: chunk positions 2T % WIN and
        // (2T + 1) % WIN, LDS plane T & 1.  Both rows' LDS reads are issued first (windows up to 15 x 15).
        auto sstep = [&](auto tc) __attribute__((always_inline)) {
            constexpr int T = decltype(tc)::value;
            constexpr int Q = (2 * T) % WIN;
            using SPc = std::integral_constant<int, T & 1>;
            using E0 = std::integral_constant<int, 0>;
            using E1 = std::integral_constant<int, 1>;
            SRow rA, rB;
            double SA[K], SB[K];
            constexpr bool BOTH = WIN <= LK_SPLIT_BOTH_MAX;                            // bigger rings leave no registers for two rows of taps
            sload(SPc{}, E0{}, rA);
            if constexpr (BOTH) sload(SPc{}, E1{}, rB);
            __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0)
#ifdef LK_ABL_S   /* timing-only ablation: results wrong */
            SA[0] = rA.q[0] * 1e-300; SA[1] = rA.sb * 1e-300; SB[0] = rB.q[0] * 1e-300; SB[1] = rB.sa * 1e-300;
            if (0)
#endif
            ssum(std::integral_constant<int, Q>{}, rA, SA);
            if constexpr (!BOTH) {
                __builtin_amdgcn_sched_barrier(0);
                sload(SPc{}, E1{}, rB);
                __builtin_amdgcn_s_waitcnt(0xC07F);
            }
#ifdef LK_ABL_S
            if (0)
#endif
            ssum(std::integral_constant<int, (Q + 1) % WIN>{}, rB, SB);
            sstore(SPc{}, E0{}, SA);
            sstore(SPc{}, E1{}, SB);
            __syncthreads();
        };
        for (int k = 0; k < 1 + C::LAG + (FW ? 1 : 0); ++k) __syncthreads();   // pipeline step 0 (FW: W / F prologue, 0..2): only the front end works
        // The van Herk chunks are anchored to ABSOLUTE image rows (chunk position of image row r = r mod WIN), so the
        // order in which a window's rows are added belongs to the output row, not to the strip: results do not depend
        // on the strip height the launcher picks from the batch size (sharded == unsharded bit for bit).
        // The unrolled body holds 2 WIN steps (every chunk position of a row pair with either LDS plane, all static);
        // the march enters it at the step T0 whose first row has chunk position kc0 and whose plane is 0:
        // 2 T0 = kc0 (mod WIN), T0 even.  One counter test per step.
        const int kc0 = ((yy0 % WIN) + WIN) % WIN;           // chunk position of marched row 0
        int t0s = (kc0 & 1) ? (kc0 + WIN) / 2 : kc0 / 2;     // 2 t0s = kc0 (mod WIN), 0 <= t0s < WIN
        if (t0s & 1) t0s += WIN;                             // WIN is odd: the other solution is even
        int cnt = -t0s;                                      // work step number of body position 0 in the first pass
        while (cnt < nsw) {
            [&]<int... Ts>(std::integer_sequence<int, Ts...>) __attribute__((always_inline)) {
                (((unsigned)cnt < (unsigned)nsw ? (sstep(std::integral_constant<int, Ts>{}), 0) : 0, ++cnt), ...);
            }(std::make_integer_sequence<int, 2 * WIN>{});
        }
        for (int k = nsw + 1 + C::LAG; k < NS; ++k) __syncthreads(); // remaining steps: only the back end works / idle tail
    }
}

}  // namespace lk
