// lk_split.h -- single-level dense LK, row-marching, WAVE-SPECIALISED: the row pipeline is cut into three roles
// and the five window-sum fields are split across the waves of a block (ParallelVideoPyr.cpp:114-191, 327-542 --
// same arithmetic as lk_march3.h).
//
// Why: the one-role kernel (lk_march3.h) keeps the (WIN-1) x 5 vertical ring of a column in ONE wave's
// registers (140 of 256 VGPRs at 15 x 15 -> 2 waves per SIMD) and sums every horizontal window out of LDS
// with R+1 reads per pixel and field.  Its ISA shows 19 ds_read2_b64 + 6 LDS stores per marched row and wave =
// ~220 LDS-pipe cycles, more than its 161 VALU instructions cost a SIMD, and a wave issues at most one
// instruction of ANY kind per 4 cycles, so its ~410 instructions per row are a >= 1640-cycle serial chain.
// Here a block of K*64 columns is marched by three kinds of waves, each with a short instruction stream:
//
//   * K/2 "F" (front) waves, TWO adjacent columns per lane: image rows (one 16-byte load of the pair + one
//     8-byte load of the left derivative tap per image and row, PF rows in flight, previous row in registers)
//     -> fx, fy, ft -> the five products of both columns -> ONE ds_write_b128 per field.
//   * 5 "S" waves, one FIELD each, K adjacent columns per lane: the lane reads the raw products of columns
//     [K*j - R, K*j + K-1 + R] as aligned 16-byte LDS reads (10 ds_read_b128 at K = 4, R = 7: 2.5 per output
//     instead of 8 reads), forms the K horizontal windows with a shared core sum (21 adds for 4 outputs at
//     15 x 15, additions only), runs the vertical van Herk / Gil-Werman ring for its K columns in registers
//     ((WIN-1) x K x 2 = 112 VGPRs at 15 x 15) and hands the K window sums back through LDS (16-byte stores).
//   * K "B" (back) waves, one column per lane: the five window sums of a row come back from LDS (one
//     ds_read_b64 per field) -> clipped-window rule, 2 x 2 solve with true fp64 divisions, /0 -> 0,
//     "+= pyrUp(coarse flow)" from register-carried coarse rows (taps fetched two rows ahead), streaming store.
//
// The three stages run concurrently on different marched rows (F: row i, S: row i-1, B: row i-2) with
// double-buffered LDS planes and ONE workgroup barrier per marched row.  Every vector-memory instruction is
// issued unconditionally (clamped rows; a store that must not happen gets an out-of-range buffer offset, which
// the hardware drops): hipcc's s_waitcnt counting gives up on loads or stores under a condition and then drains
// vmcnt(0) at every use, which serialises a step behind its own prefetch.
// K = 4 up to 15 x 15 (11 waves, <= 168 VGPRs: 3 waves per SIMD); K = 2 for bigger windows (8 waves), whose
// ring would not fit a wave at K = 4.
//
// Summation order: horizontal = tree over the shared core + short left/right chains, vertical = van Herk; only
// additions (flat regions give exactly 0 like the reference); same products, solve and pyrUp expressions as
// lk_march3.h.  Host-checked: W even, H > R + PF + 1, even strip height (row parity is static in the unrolled
// pipeline); other shapes take lk_march3.h.
#pragma once
#include "lk_w3x2.h"     // buf_ld2 (16-byte buffer loads), lk_march3.h

#ifndef LK_SPLIT_PF
#define LK_SPLIT_PF 2
#endif

#ifdef LK_SPLIT_STAMPS   /* diagnostic build: per-wave segment cycle totals -> a.dbg (tools/split_stamps_summary.py) */
#define SP_STAMP(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#define SP_ACC(k, a, b) seg[k] += (b) - (a)
#else
#define SP_STAMP(var)
#define SP_ACC(k, a, b)
#endif

namespace lk {

typedef double lk_d2 __attribute__((ext_vector_type(2)));

template <int WIN, int K>
struct SplitCfg {
    static constexpr int R = WIN / 2;
    static constexpr int PF = LK_SPLIT_PF;           // image rows in flight per F lane: a pipeline step is short, HBM latency is not
    static constexpr int NB = K;                     // B waves: 64 columns each
    static constexpr int NF = K / 2;                 // F waves: 128 columns each
    static constexpr int TXC = 64 * K;               // columns per block
    static constexpr int HL = R + (R & 1);           // halo columns on each side (even: a lane's column pair is 16-byte aligned)
    static constexpr int XO = TXC - 2 * HL;          // output columns per block
    static constexpr int NW = NB + 5 + NF;           // waves per block, in this order: B, S, F
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
#ifdef LK_SPLIT_WPE
    static constexpr int WPE = LK_SPLIT_WPE;
#else
    static constexpr int WPE = (K == 2 && WIN <= 21) ? 4 : 1;   // waves per SIMD to compile for (8-wave blocks: 4 -> two blocks per CU)
#endif
    static constexpr int UN = 2;                     // unrolled F group (LDS buffer parity, queue slot)
    static_assert(K == 2 || K == 4, "S lanes own 2 or 4 columns");
    static_assert(PSTR >= NJP, "product half-planes must not overlap");
    static_assert(UN % PF == 0, "static queue slots: PF is 1 or 2");
};

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

template <int WIN, int K, bool HAS_UP>
__global__ __launch_bounds__((SplitCfg<WIN, K>::NT), (SplitCfg<WIN, K>::WPE)) void lk_split_kernel(LKArgs a)
{
    using C = SplitCfg<WIN, K>;
    constexpr int R = C::R, TXC = C::TXC, XO = C::XO, KH = C::KH, PADJ = C::PADJ, PF = C::PF, UN = C::UN, HL = C::HL;
    __shared__ __attribute__((aligned(16))) double s_prod[2][5][C::PROD_D];
    __shared__ __attribute__((aligned(16))) double s_sums[2][5][C::SUMS_D];

    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
#ifdef LK_SPLIT_STAMPS
    unsigned long long seg[6] = {0, 0, 0, 0, 0, 0};
#endif
    const int pair = blockIdx.z;
    const int H = a.H, W = a.W;
    const int y0 = blockIdx.y * a.th;                // even (host)
    const int y1 = min(y0 + a.th, H);
    const int nm = (y1 - y0) + 2 * R;                // marched product rows: image rows y0-R .. y1-1+R
    const int yy0 = y0 - R;
    const int x0 = (int)blockIdx.x * XO - HL;        // image column of block column 0 (even)
    const int nit = 2 * R + 2 + (y1 - y0 + 3) / 4 * 4;   // pipeline steps (= barriers): fill + output rows padded to whole groups of 4

    // zero the pad lanes of the product planes once (only non-output columns ever read them)
    for (int k = threadIdx.x; k < 2 * 5 * KH * 2 * PADJ; k += C::NT) {
        const int side = k % 2, rest = k / 2;
        const int pj = rest % PADJ, h = (rest / PADJ) % KH, f = (rest / (PADJ * KH)) % 5, b = rest / (PADJ * KH * 5);
        const int chunk = h * C::PSTR + (side ? PADJ + 64 + pj : pj);
        s_prod[b][f][2 * chunk] = 0.0;
        s_prod[b][f][2 * chunk + 1] = 0.0;
    }
    __syncthreads();

    if (wv >= C::NB + 5) {
        // =========================================================================================
        // F waves: two adjacent columns per lane -> products
        // =========================================================================================
        const int m = (int)threadIdx.x - 64 * (C::NB + 5);   // column pair of the block: columns 2m, 2m + 1
        const int xa = x0 + 2 * m;                   // even; W even: the pair is all in or all out
        const int pitchb = (int)a.i1.pitch * 8;
        const bool col_in = xa >= 0 && xa < W;
        const int xc = min(max(xa, 0), W - 2);
        const int row_lo = max(y0 - R - PF, 0);
        const long long img_bytes = ((long long)(H - 1 - row_lo) * a.i1.pitch + W) * 8;
        const __amdgpu_buffer_rsrc_t rI1 = buf_rsrc(a.i1.at(pair) + (long long)row_lo * a.i1.pitch, img_bytes);
        const __amdgpu_buffer_rsrc_t rI2 = buf_rsrc(a.i2.at(pair) + (long long)row_lo * a.i2.pitch, img_bytes);
        const int vxc = xc * 8;
        const int vxl = (xc > 0 ? xc - 1 : 1) * 8;   // left derivative tap of column xa (reflect-101 at column 0)
        const int p_chunk = KH == 1 ? PADJ + m : (m & 1) * C::PSTR + PADJ + (m >> 1);

        // byte offset of image row yy: one reflection (exact for every row an output needs: H > R + PF + 1),
        // then clamped (prefetches of the idle tail steps run past the last needed row)
        auto img_row_off = [&](int yy) {
            int rr = yy < 0 ? -yy : yy;
            rr = min(rr, 2 * H - 2 - rr);
            rr = max(rr, row_lo);
            return (rr - row_lo) * pitchb;
        };
        // previous row (p*) and current row (c*): left tap l, columns a and b; [0] image 1, [1] image 2
        double pl[2], pa[2], pb[2], cl[2], ca[2], cb[2];
        {
            int ro = img_row_off(yy0 - 1);
            pl[0] = buf_ld(rI1, vxl, ro); buf_ld2(rI1, vxc, ro, pa[0], pb[0]);
            pl[1] = buf_ld(rI2, vxl, ro); buf_ld2(rI2, vxc, ro, pa[1], pb[1]);
            ro = img_row_off(yy0);
            cl[0] = buf_ld(rI1, vxl, ro); buf_ld2(rI1, vxc, ro, ca[0], cb[0]);
            cl[1] = buf_ld(rI2, vxl, ro); buf_ld2(rI2, vxc, ro, ca[1], cb[1]);
        }
        double ql[PF][2], qa[PF][2], qb[PF][2];              // queue slot of marched row i = i % PF
#pragma unroll
        for (int r = 1; r < PF; ++r) {
            const int o = img_row_off(yy0 + r);
            ql[r][0] = buf_ld(rI1, vxl, o); buf_ld2(rI1, vxc, o, qa[r][0], qb[r][0]);
            ql[r][1] = buf_ld(rI2, vxl, o); buf_ld2(rI2, vxc, o, qa[r][1], qb[r][1]);
        }
        ql[0][0] = ql[0][1] = qa[0][0] = qa[0][1] = qb[0][0] = qb[0][1] = 0.0;

        auto fstep = [&](auto uc, int it) __attribute__((always_inline)) {
            constexpr int U = decltype(uc)::value;
            constexpr int PAR = U & 1;
            SP_STAMP(t0);
            {                                               // prefetch image row it + PF into the slot freed by the last rotate
                const int rn = img_row_off(yy0 + it + PF);
                ql[U % PF][0] = buf_ld(rI1, vxl, rn); buf_ld2(rI1, vxc, rn, qa[U % PF][0], qb[U % PF][0]);
                ql[U % PF][1] = buf_ld(rI2, vxl, rn); buf_ld2(rI2, vxc, rn, qa[U % PF][1], qb[U % PF][1]);
            }
            const int yy = yy0 + it;
            const bool keep = yy >= 0 && yy < H && col_in;
            double fxa = f2d_x(pl[0], pa[0], cl[0], ca[0]) + f2d_x(pl[1], pa[1], cl[1], ca[1]);
            double fya = f2d_y(pl[0], pa[0], cl[0], ca[0]) + f2d_y(pl[1], pa[1], cl[1], ca[1]);
            const double fta = f2d_neg(pl[0], pa[0], cl[0], ca[0]) + f2d_pos(pl[1], pa[1], cl[1], ca[1]);
            double fxb = f2d_x(pa[0], pb[0], ca[0], cb[0]) + f2d_x(pa[1], pb[1], ca[1], cb[1]);
            double fyb = f2d_y(pa[0], pb[0], ca[0], cb[0]) + f2d_y(pa[1], pb[1], ca[1], cb[1]);
            const double ftb = f2d_neg(pa[0], pb[0], ca[0], cb[0]) + f2d_pos(pa[1], pb[1], ca[1], cb[1]);
            fxa = keep ? fxa : 0.0; fya = keep ? fya : 0.0;
            fxb = keep ? fxb : 0.0; fyb = keep ? fyb : 0.0;
            lk_d2 o;
            o.x = fxa * fxa; o.y = fxb * fxb; *reinterpret_cast<lk_d2 *>(&s_prod[PAR][0][2 * p_chunk]) = o;
            o.x = fya * fya; o.y = fyb * fyb; *reinterpret_cast<lk_d2 *>(&s_prod[PAR][1][2 * p_chunk]) = o;
            o.x = fxa * fya; o.y = fxb * fyb; *reinterpret_cast<lk_d2 *>(&s_prod[PAR][2][2 * p_chunk]) = o;
            o.x = fxa * fta; o.y = fxb * ftb; *reinterpret_cast<lk_d2 *>(&s_prod[PAR][3][2 * p_chunk]) = o;
            o.x = fya * fta; o.y = fyb * ftb; *reinterpret_cast<lk_d2 *>(&s_prod[PAR][4][2 * p_chunk]) = o;
            SP_STAMP(t1);
            // rotate: the only wait for the prefetch
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                pl[g] = cl[g]; pa[g] = ca[g]; pb[g] = cb[g];
                cl[g] = ql[(U + 1) % PF][g]; ca[g] = qa[(U + 1) % PF][g]; cb[g] = qb[(U + 1) % PF][g];
            }
#ifdef LK_SPLIT_STAMPS
            asm volatile("" :: "v"(cl[0]), "v"(ca[0]), "v"(cb[0]), "v"(cl[1]), "v"(ca[1]), "v"(cb[1]));
#endif
            SP_STAMP(t2);
            __syncthreads();
            SP_STAMP(t3);
            SP_ACC(0, t0, t1); SP_ACC(2, t1, t2); SP_ACC(3, t2, t3);
        };
        for (int base = 0; base < nit; base += UN) {         // nit is a multiple of UN: no guards inside a group
            [&]<int... Us>(std::integer_sequence<int, Us...>) __attribute__((always_inline)) {
                (fstep(std::integral_constant<int, Us>{}, base + Us), ...);
            }(std::make_integer_sequence<int, UN>{});
        }
    } else if (wv < C::NB) {
        // =========================================================================================
        // B waves: one column per lane: window sums -> flow
        // =========================================================================================
        const int t = threadIdx.x;                   // block column 0 .. TXC-1
        const int x = x0 + t;
        const int xc = min(max(x, 0), W - 1);
        const bool is_out = t >= HL && t < HL + XO && x < W;
        const bool inner_x = x >= 1 && x <= W - 2;
        const bool clipped_x = (x - R < 0) || (x + R > W - 1);
        const bool fix_x = is_out && inner_x && clipped_x;      // lanes whose window is clipped horizontally (first / last block)
        const bool any_fix_x = __builtin_amdgcn_ballot_w64(fix_x) != 0;
        const int s_slot = KH == 1 ? t : 2 * (((t % K) >> 1) * C::SSTR + t / K) + (t & 1);
        const int nrows = y1 - y0;

        [[maybe_unused]] UpSel1 us{};
        [[maybe_unused]] double hu[3] = {0, 0, 0}, hv[3] = {0, 0, 0};   // carried coarse rows (rotated every even row)
        [[maybe_unused]] double tp[2][6];                               // taps of the next coarse rows (u: 0..2, v: 3..5), two in flight
#pragma unroll
        for (int k = 0; k < 6; ++k) tp[0][k] = tp[1][k] = 0.0;
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
            int su, sv;
            tap_off(y0 - 1, su, sv);                            // taps the first odd row (y0 + 1) consumes -> buffer 0
            tp[0][0] = buf_ld(rCU, us.jm, su); tp[0][1] = buf_ld(rCU, us.jc, su); tp[0][2] = buf_ld(rCU, us.jp, su);
            tp[0][3] = buf_ld(rCV, us.jm, sv); tp[0][4] = buf_ld(rCV, us.jc, sv); tp[0][5] = buf_ld(rCV, us.jp, sv);
        }
        const int vxo = x * 8;
        int sou = 0, sov = 0;                                   // byte offsets of output row r in the u / v strips

        // One output row r = y - y0 (< nrows) at static position U of the unrolled group (y0 is even: row parity =
        // U & 1 = parity of the pipeline step = LDS buffer; tap buffer = (U >> 1) & 1).  FAST: the window is
        // vertically inside the image, so nothing but the per-lane column masks is conditional.  Both variants
        // issue the same vector-memory instructions, unconditionally (see the file header).
        const int vo_lane = is_out ? vxo : (int)0x80000000;   // out of range: the store is dropped
        auto brow = [&](auto uc, auto fastc, int r) __attribute__((always_inline)) {
            constexpr int U = decltype(uc)::value;
            constexpr bool FAST = decltype(fastc)::value;
            constexpr int PAR = U & 1;
            constexpr int TB = (U >> 1) & 1;
            const int y = y0 + r;
            if constexpr (HAS_UP && PAR == 1) {             // taps consumed by the NEXT odd row (y + 2)
                int su, sv;
                tap_off(y, su, sv);
                tp[TB ^ 1][0] = buf_ld(rCU, us.jm, su); tp[TB ^ 1][1] = buf_ld(rCU, us.jc, su); tp[TB ^ 1][2] = buf_ld(rCU, us.jp, su);
                tp[TB ^ 1][3] = buf_ld(rCV, us.jm, sv); tp[TB ^ 1][4] = buf_ld(rCV, us.jc, sv); tp[TB ^ 1][5] = buf_ld(rCV, us.jp, sv);
            }
            double S[5];
#pragma unroll
            for (int f = 0; f < 5; ++f) S[f] = s_sums[PAR][f][s_slot];
            bool inner, fix;
            if constexpr (FAST) {
                inner = inner_x;
                fix = any_fix_x && fix_x;
            } else {
                inner = y >= 1 && y <= H - 2 && inner_x;
                fix = is_out && inner && ((y - R < 0) || (y + R > H - 1) || clipped_x);
            }
            if (fix) {                                      // SerialPyr.cpp:94-95: partial window -> m(y,x) * WIN^2
                double pc[5];
                lk_products_at(a.i1.at(pair), a.i2.at(pair), a.i1.pitch, H, W, y, x, pc);
#pragma unroll
                for (int f = 0; f < 5; ++f) S[f] = pc[f] * (double)(WIN * WIN);
            }
            const double det = S[0] * S[1] - S[2] * S[2];
            const double un = S[2] * S[4] - S[1] * S[3];
            const double vn = S[3] * S[2] - S[0] * S[4];
            const bool nz = det != 0.0 && inner;            // cv::divide: x/0 -> 0; outermost ring: 0
            const double qu = un / det, qv = vn / det;
            double u = nz ? qu : 0.0;
            double v = nz ? qv : 0.0;
            if constexpr (HAS_UP) {
                if constexpr (PAR == 0) {                   // rotate in the coarse row turned into slot 0 last row
                    const double nu = hu[0], nv = hv[0];
                    hu[0] = hu[1]; hu[1] = hu[2]; hu[2] = nu;
                    hv[0] = hv[1]; hv[1] = hv[2]; hv[2] = nv;
                    u += (hu[0] + hu[1] * 6 + hu[2]) * (1.0 / 64);
                    v += (hv[0] + hv[1] * 6 + hv[2]) * (1.0 / 64);
                } else {
                    u += ((hu[1] + hu[2]) * 4) * (1.0 / 64);
                    v += ((hv[1] + hv[2]) * 4) * (1.0 / 64);
                    hu[0] = pyrup_h1v(tp[TB][0], tp[TB][1], tp[TB][2], us);   // next coarse row, used from row y+1 on
                    hv[0] = pyrup_h1v(tp[TB][3], tp[TB][4], tp[TB][5], us);
                }
            }
            buf_st(rU, vo_lane, sou, u);
            buf_st(rV, vo_lane, sov, v);
            sou += upb; sov += vpb;
        };
        auto bstep = [&](auto uc, int r) __attribute__((always_inline)) {
            SP_STAMP(t0);
            if (H > 2 * R && (unsigned)(y0 + r - R) <= (unsigned)(H - 1 - 2 * R)) brow(uc, std::true_type{}, r);
            else brow(uc, std::false_type{}, r);
            SP_STAMP(t1);
            __syncthreads();
            SP_STAMP(t2);
            SP_ACC(1, t0, t1); SP_ACC(3, t1, t2);
        };
        for (int k = 0; k < 2 * R + 2; ++k) __syncthreads();   // pipeline fill: no window sums yet
        int r = 0;
        for (; r + 4 <= nrows; r += 4) {                        // whole groups
            bstep(std::integral_constant<int, 0>{}, r);
            bstep(std::integral_constant<int, 1>{}, r + 1);
            bstep(std::integral_constant<int, 2>{}, r + 2);
            bstep(std::integral_constant<int, 3>{}, r + 3);
        }
        if (r < nrows) {                                        // nrows is even: a last half group, then two idle steps
            bstep(std::integral_constant<int, 0>{}, r);
            bstep(std::integral_constant<int, 1>{}, r + 1);
            __syncthreads();
            __syncthreads();
        }
    } else {
        // =========================================================================================
        // S waves: one field each, K adjacent columns per lane
        // =========================================================================================
        const int f = wv - C::NB;
        const int j = threadIdx.x & 63;
        constexpr int NV = K * (2 * PADJ + 1);               // raw values a lane can see: offsets -K*PADJ .. K*PADJ+K-1
        constexpr int O0 = K * PADJ;                         // index of column offset 0
        double ring[WIN][K], pre[K];
#pragma unroll
        for (int q = 0; q < WIN; ++q)
#pragma unroll
            for (int c = 0; c < K; ++c) ring[q][c] = 0.0;
#pragma unroll
        for (int c = 0; c < K; ++c) pre[c] = 0.0;

        auto srow = [&](auto kc, int i) __attribute__((always_inline)) {
            constexpr int KC = decltype(kc)::value;
            SP_STAMP(t0);
            const double *pp = &s_prod[i & 1][f][0];
            double v[NV];
#pragma unroll
            for (int d = -PADJ; d <= PADJ; ++d)
#pragma unroll
                for (int h = 0; h < KH; ++h) {
                    constexpr int dummy = 0; (void)dummy;
                    const int o0 = K * d + 2 * h;            // column offsets o0, o0 + 1
                    if (o0 + 1 >= -R && o0 <= K - 1 + R) {
                        const lk_d2 c2 = *reinterpret_cast<const lk_d2 *>(pp + 2 * (h * C::PSTR + PADJ + j + d));
                        v[O0 + o0] = c2.x; v[O0 + o0 + 1] = c2.y;
                    } else {
                        v[O0 + o0] = 0.0; v[O0 + o0 + 1] = 0.0;
                    }
                }
            // horizontal windows of the K columns: shared core [K-1-R, R] + left / right extensions
            const double core = split_tree<O0 + K - 1 - R, O0 + R>(v);
            double hs[K];
            {
                double L[K], Rr[K];                          // L[m] = sum of the m columns left of the core, Rr[m] likewise right
                L[0] = 0.0; Rr[0] = 0.0;
#pragma unroll
                for (int m = 1; m < K; ++m) {
                    L[m] = (m == 1) ? v[O0 + K - 1 - R - 1] : L[m - 1] + v[O0 + K - 1 - R - m];
                    Rr[m] = (m == 1) ? v[O0 + R + 1] : Rr[m - 1] + v[O0 + R + m];
                }
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    double s = core;
                    if (K - 1 - c > 0) s += L[K - 1 - c];
                    if (c > 0) s += Rr[c];
                    hs[c] = s;
                }
            }
#ifdef LK_SPLIT_STAMPS
#pragma unroll
            for (int c = 0; c < K; ++c) asm volatile("" :: "v"(hs[c]));
#endif
            SP_STAMP(t1);
            // vertical van Herk / Gil-Werman step at static chunk position KC
            double S[K];
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
            if (i >= 2 * R) {                                // a full vertical window: hand the K sums to the back end
                double *sp = &s_sums[i & 1][f][0];
#pragma unroll
                for (int h = 0; h < KH; ++h) {
                    lk_d2 o; o.x = S[2 * h]; o.y = S[2 * h + 1];
                    *reinterpret_cast<lk_d2 *>(sp + 2 * (h * C::SSTR + j)) = o;
                }
            }
            SP_STAMP(t2);
            __syncthreads();
            SP_STAMP(t3);
            SP_ACC(0, t0, t1); SP_ACC(1, t1, t2); SP_ACC(3, t2, t3);
        };
        __syncthreads();                                     // pipeline step 0: only the front end works
        for (int base = 0; base < nm; base += WIN) {
            [&]<int... Ks>(std::integer_sequence<int, Ks...>) __attribute__((always_inline)) {
                ((base + Ks < nm ? (srow(std::integral_constant<int, Ks>{}, base + Ks), 0) : 0), ...);
            }(std::make_integer_sequence<int, WIN>{});
        }
        for (int k = nm + 1; k < nit; ++k) __syncthreads(); // pipeline steps nm + 1 ..: only the back end works / idle tail
    }
#ifdef LK_SPLIT_STAMPS
    if (a.dbg && (threadIdx.x & 63) == 0) {
        const unsigned wid = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * C::NW + wv;
        for (int k = 0; k < 6; ++k) a.dbg[wid * 8 + k] = seg[k];
        a.dbg[wid * 8 + 6] = nit;
        a.dbg[wid * 8 + 7] = wv;
    }
#endif
}

}  // namespace lk
