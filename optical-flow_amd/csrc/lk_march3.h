// lk_march3.h -- single-level dense LK, row-marching kernel, one column per thread, lean body.
//
// What bounds this kernel is fp64 VALU / LDS / VMEM instruction issue at the occupancy the register
// ring allows, so the body is organised for instruction count and for <= 256 VGPRs (2 waves per SIMD
// up to WIN = 15):
//   * thread t owns column x = x_base + t; per row: one 8-byte buffer load per image + one for the left
//     derivative tap (reflect-101 folded into the index; scalar row offset, no vector address
//     arithmetic), previous row carried in registers, PF rows prefetched (PF = 3 for WIN = 3, else 1);
//   * horizontal WIN-tap sums: lanes 2j,2j+1 form the pair sum q_j = p(2j)+p(2j+1) with one DPP
//     quad-permute (no LDS round trip); every lane stores p, even lanes store q; a window is then
//     R pair sums + one single = R+1 LDS reads and R adds per field instead of WIN and WIN-1;
//     ONE barrier per marched row (double-buffered LDS rows).  WIN = 3 in single-wave blocks (TX = 64):
//     the window is two DPP wave shifts -- no LDS, no barrier;
//   * vertical WIN-tap sums: van Herk / Gil-Werman prefix + suffix ring in REGISTERS, statically
//     indexed (the chunk of WIN rows is unrolled through a template fold); additions only -- no
//     running subtract, so flat regions stay exactly 0 like the reference;
//   * "+= pyrUp(coarse flow)": the horizontally filtered coarse rows are carried in registers and
//     refreshed every second fine row; their taps are fetched 3 per row (WIN divisible by 3) or 6.
#pragma once
#include "lk_kernels.h"

namespace lk {

// swap with the neighbouring lane (lane ^ 1) through DPP quad_perm [1,0,3,2]
__device__ __forceinline__ double dpp_swap1(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int slo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
    const int shi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
    return __hiloint2double(shi, slo);
}

// Horizontal cv::pyrUp tap of one coarse row for ONE fine column, as three per-lane weights:
//   h = (A*wA + B*wB) + C*wC   with A = c[j-1], B = c[j], C = c[j+1] (clamped indices)
//   even x: interior (1,6,1) | left (0,6,2) | right (1,7,0) | single (0,8,0)
//   odd  x: interior/left (0,4,4) | right/single (0,8,0)
// Every weight on A and C is 0 or a power of two, so A*wA and C*wC are exact and the two FMAs round
// exactly where pyrup_h()'s additions round ((B+C)*4 == B*4 + C*4 bit for bit).
struct UpSel1 {
    int jm, jc, jp;         // BYTE offsets of the three taps inside a coarse row
    double wA, wB, wC;
};
__device__ __forceinline__ UpSel1 make_upsel1(int x, int Wc)
{
    UpSel1 s;
    int j = x >> 1;
    j = j < 0 ? 0 : (j > Wc - 1 ? Wc - 1 : j);
    s.jc = j * 8;
    s.jm = (j > 0 ? j - 1 : 0) * 8;
    s.jp = (j < Wc - 1 ? j + 1 : Wc - 1) * 8;
    const bool single = Wc == 1, left = j == 0, right = j == Wc - 1, odd = (x & 1) != 0;
    if (odd) {
        s.wA = 0.0;
        s.wB = (single || right) ? 8.0 : 4.0;
        s.wC = (single || right) ? 0.0 : 4.0;
    } else {
        s.wA = (single || left) ? 0.0 : 1.0;
        s.wB = single ? 8.0 : (right ? 7.0 : 6.0);
        s.wC = (single || right) ? 0.0 : (left ? 2.0 : 1.0);
    }
    return s;
}
// A tap with weight 0 is a tap cv::pyrUp does not read: it is dropped BEFORE the multiplication, so that a non-finite
// coarse flow value (a division by a denormal determinant) spreads exactly as far as in the reference (Inf * 0 would
// turn the neighbouring columns into NaN).  The comparisons are loop-invariant per lane.
__device__ __forceinline__ double pyrup_h1v(double A, double B, double C, const UpSel1 &s)
{
    const double a = s.wA != 0.0 ? A : 0.0, c = s.wC != 0.0 ? C : 0.0;
    return __builtin_fma(c, s.wC, __builtin_fma(a, s.wA, B * s.wB));
}

template <int WIN, int TX, bool HAS_UP>
__global__ __launch_bounds__(TX, (WIN <= 15 ? 2 : 1)) void lk_march3_kernel(LKArgs a)
{
    constexpr int R = WIN / 2;
    constexpr int XO = TX - 2 * R;            // output columns per block
    constexpr int PP = 16, PQ = 8;            // LDS pads (only non-output threads read them)
    // WIN == 3 in single-wave blocks: the whole horizontal window is reachable by DPP (62 of 64 lanes produce output)
    constexpr bool DPP_ONLY = (WIN == 3 && TX == 64);
    __shared__ double s_p[DPP_ONLY ? 1 : 2][DPP_ONLY ? 1 : 5][DPP_ONLY ? 1 : TX + 2 * PP];
    __shared__ double s_q[DPP_ONLY ? 1 : 2][DPP_ONLY ? 1 : 5][DPP_ONLY ? 1 : TX / 2 + 2 * PQ];

    const int t = threadIdx.x;
    const int pair = blockIdx.z;
    const int H = a.H, W = a.W;
    const int x = (int)blockIdx.x * XO - R + t;
    const int y0 = blockIdx.y * a.th;
    const int y1 = min(y0 + a.th, H);
    const int pitch = (int)a.i1.pitch;       // one plane is < 2^31 elements (checked on the host)
    const bool col_in = x >= 0 && x < W;
    const int xc = min(max(x, 0), W - 1);
    const int xl = r101(xc - 1, W);
    // Prefetch distance in marched rows.  A row of a small window is short (w3: ~0.25 us of work), so one row
    // of lookahead does not cover the HBM latency: keep PF rows of image taps (and of coarse taps) in flight.
    // PF must divide WIN so that queue slots are static in the unrolled chunk.  Big windows: 1 (registers).
    constexpr int PF = (WIN == 3) ? 3 : 1;
    static_assert(WIN % PF == 0, "queue slots are indexed by the static chunk position");
    // buffer descriptors based at the first row the strip can touch (so that 32-bit offsets suffice for any
    // plane size): rows [y0-R-1, y1-1+R+PF] reflected once into the image (tiny images: anywhere); per-thread
    // byte offsets of the own column and of the left derivative tap
    const int row_lo = (H <= R + PF) ? 0 : max(y0 - R - PF, 0);
    const long long img_bytes = ((long long)(H - 1 - row_lo) * pitch + W) * 8;
    const __amdgpu_buffer_rsrc_t rI1 = buf_rsrc(a.i1.at(pair) + (long long)row_lo * pitch, img_bytes);
    const __amdgpu_buffer_rsrc_t rI2 = buf_rsrc(a.i2.at(pair) + (long long)row_lo * pitch, img_bytes);
    const int vxc = xc * 8, vxl = xl * 8;
    const int pitchb = pitch * 8;
    const bool is_out = t >= R && t < TX - R && x < W;

    // window decomposition in thread-index space: threads [t-R, t+R] = R pairs + one single
    const bool lead_single = ((t - R) & 1) != 0;               // (t-R) odd: single first
    const int jb = (t - R + 1) >> 1;                           // first full pair (arithmetic shift)
    const int sidx = lead_single ? t - R : t + R;              // the single tap

    const int nmarch = (y1 - y0) + 2 * R;
    const int yy0 = y0 - R;

    double ring[WIN][5];
#pragma unroll
    for (int q = 0; q < WIN; ++q)
#pragma unroll
        for (int f = 0; f < 5; ++f) ring[q][f] = 0.0;
    double pre[5] = {0, 0, 0, 0, 0};
    [[maybe_unused]] Sum9 win9[5];                             // WIN == 3: partial window sums in the reference's order (lk_device.h)
#pragma unroll
    for (int f = 0; f < 5; ++f) win9[f].a = win9[f].b = 0.0;

    // byte offset of image row yy (reflected into the image; apron rows carry the reflect-101 row, so that at
    // yy = 0 the "previous" row is row 1 = r101(-1); one reflection suffices unless the image is shorter than
    // the apron)
    auto img_row_off = [&](int yy) {
        int rr = yy < 0 ? -yy : yy;
        rr = rr >= H ? 2 * H - 2 - rr : rr;
        if (H <= R + PF) rr = r101(yy, H);
        return (rr - row_lo) * pitchb;
    };
    // previous row (pa*, pb*), current row (ca*, cb*) and the queue of rows in flight: [0] = left tap, [1] = own
    int ro = (r101(yy0 - 1, H) - row_lo) * pitchb;
    double pa0 = buf_ld(rI1, vxl, ro), pa1 = buf_ld(rI1, vxc, ro), pb0 = buf_ld(rI2, vxl, ro), pb1 = buf_ld(rI2, vxc, ro);
    ro = (r101(yy0, H) - row_lo) * pitchb;
    double ca0 = buf_ld(rI1, vxl, ro), ca1 = buf_ld(rI1, vxc, ro), cb0 = buf_ld(rI2, vxl, ro), cb1 = buf_ld(rI2, vxc, ro);
    double qa0[PF], qa1[PF], qb0[PF], qb1[PF];                  // slot of marched row i = i % PF
#pragma unroll
    for (int r = 1; r < PF; ++r) {
        const int o = img_row_off(yy0 + r);
        qa0[r] = buf_ld(rI1, vxl, o); qa1[r] = buf_ld(rI1, vxc, o); qb0[r] = buf_ld(rI2, vxl, o); qb1[r] = buf_ld(rI2, vxc, o);
    }
    qa0[0] = qa1[0] = qb0[0] = qb1[0] = 0.0;

    [[maybe_unused]] UpSel1 us{};
    [[maybe_unused]] double hu[3] = {0, 0, 0}, hv[3] = {0, 0, 0};   // carried coarse rows s(i-1),s(i),s(i+1)
    [[maybe_unused]] const int Hc = H >> 1;
    [[maybe_unused]] const int cpub = (int)a.cu.pitch * 8, cpvb = (int)a.cv.pitch * 8;
    [[maybe_unused]] const int crow_lo = max((y0 >> 1) - 1, 0);
    auto up_srow = [&](int k) { return k < 0 ? min(1, Hc - 1) : (k > Hc - 1 ? Hc - 1 : k); };
    // (without HAS_UP the coarse descriptors are never used; they alias image 1 so that they are well-formed)
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rCU =
        HAS_UP ? buf_rsrc(a.cu.at(pair) + (long long)crow_lo * a.cu.pitch, ((long long)(Hc - 1 - crow_lo) * a.cu.pitch + (W >> 1)) * 8) : rI1;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rCV =
        HAS_UP ? buf_rsrc(a.cv.at(pair) + (long long)crow_lo * a.cv.pitch, ((long long)(Hc - 1 - crow_lo) * a.cv.pitch + (W >> 1)) * 8) : rI2;
    if (HAS_UP) {
        us = make_upsel1(xc, W >> 1);
        const int i0 = y0 >> 1;                                 // slots of the first output row
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int sr = up_srow(i0 - 1 + s) - crow_lo;
            hu[s] = pyrup_h1v(buf_ld(rCU, us.jm, sr * cpub), buf_ld(rCU, us.jc, sr * cpub), buf_ld(rCU, us.jp, sr * cpub), us);
            hv[s] = pyrup_h1v(buf_ld(rCV, us.jm, sr * cpvb), buf_ld(rCV, us.jc, sr * cpvb), buf_ld(rCV, us.jp, sr * cpvb), us);
        }
    }
    // coarse taps of the "+= pyrUp" term, queued like the image rows: slot of marched row i = i % PF;
    // windows divisible by 3 use the 3-taps-per-row scheme (see the row body) with a 3-slot queue
    constexpr bool TAP3 = HAS_UP && (WIN % 3 == 0);
    constexpr int LA = PF > 1 ? 1 : 0;                          // TAP3 lookahead in rows (the first output row is >= 2 rows in)
    [[maybe_unused]] double tq3[3][3];
    [[maybe_unused]] double tq[PF][6];
    auto tap_row = [&](int y) { return max(up_srow(((y + 1) >> 1) + 1) - crow_lo, 0); };   // apron rows: clamped, unused
    if (HAS_UP && !TAP3) {
#pragma unroll
        for (int r = 0; r < PF - 1; ++r) {
            const int sr = tap_row(yy0 - R + r);
            tq[r][0] = buf_ld(rCU, us.jm, sr * cpub); tq[r][1] = buf_ld(rCU, us.jc, sr * cpub); tq[r][2] = buf_ld(rCU, us.jp, sr * cpub);
            tq[r][3] = buf_ld(rCV, us.jm, sr * cpvb); tq[r][4] = buf_ld(rCV, us.jc, sr * cpvb); tq[r][5] = buf_ld(rCV, us.jp, sr * cpvb);
        }
    }
    // outputs: descriptor based at the tile's first output row
    const int upb = (int)a.u.pitch * 8, vpb = (int)a.v.pitch * 8;
    const __amdgpu_buffer_rsrc_t rU = buf_rsrc(a.u.at(pair) + (long long)y0 * a.u.pitch, ((long long)(y1 - 1 - y0) * a.u.pitch + W) * 8);
    const __amdgpu_buffer_rsrc_t rV = buf_rsrc(a.v.at(pair) + (long long)y0 * a.v.pitch, ((long long)(y1 - 1 - y0) * a.v.pitch + W) * 8);
    const int vxo = x * 8;

    // One marched row at chunk position K (compile-time): the chunk loop below is unrolled WIN
    // times so that every ring access is a plain register (no dispatch, no copies).
#define LK3_STAMP(var)
    auto row = [&](auto kc, int i) __attribute__((always_inline)) {
        constexpr int K = decltype(kc)::value;
        LK3_STAMP(t0);
        const int bsel = i & 1;
        const int yy = yy0 + i;
        const bool row_in = yy >= 0 && yy < H;
        // prefetch image row yy + PF into the queue slot that the rotate of the previous iteration freed
        // (unconditional; consumed by a rotate PF iterations later, so its latency hides behind PF rows of work)
        {
            const int rn = img_row_off(yy + PF);
            qa0[K % PF] = buf_ld(rI1, vxl, rn); qa1[K % PF] = buf_ld(rI1, vxc, rn);
            qb0[K % PF] = buf_ld(rI2, vxl, rn); qb1[K % PF] = buf_ld(rI2, vxc, rn);
        }
        const int y = yy - R;                                   // output row of this iteration
        const bool out_row = i >= 2 * R;
        // the NEXT coarse row of the "+= pyrUp" term is turned into the new carried slot in the epilogue of
        // odd output rows (slot 0 is dead there).  Its three taps per plane are fetched at the top of EVERY
        // row, unconditionally: with a conditional fetch the compiler's s_waitcnt before the products
        // (which only need the row prefetched one iteration ago) counts the minimum over both paths and
        // ends up waiting for the image loads issued just above on every odd row.
        if constexpr (TAP3) {
            // three taps per row instead of six: the taps are consumed only in the epilogue of ODD output rows, so an
            // even row fetches the u-plane taps and the odd row after it the v-plane taps of the same coarse row;
            // plane descriptor and row offset are selected with scalar ops (the row parity is wave-uniform), the
            // loads stay unconditional.  Slot of output row yq = its chunk position % 3.
            const int yq = y + LA;
            const bool odd = (yq & 1) != 0;
            const __amdgpu_buffer_rsrc_t rs = odd ? rCV : rCU;
            const int so = tap_row(yq | 1) * (odd ? cpvb : cpub);
            constexpr int ts = (K + LA) % 3;
            tq3[ts][0] = buf_ld(rs, us.jm, so); tq3[ts][1] = buf_ld(rs, us.jc, so); tq3[ts][2] = buf_ld(rs, us.jp, so);
        } else if (HAS_UP) {                                    // taps of output row y + PF - 1 -> slot (K + PF - 1) % PF
            constexpr int ts = (K + PF - 1) % PF;
            const int sr = tap_row(y + PF - 1);
            const int su = sr * cpub, sv = sr * cpvb;
            tq[ts][0] = buf_ld(rCU, us.jm, su); tq[ts][1] = buf_ld(rCU, us.jc, su); tq[ts][2] = buf_ld(rCU, us.jp, su);
            tq[ts][3] = buf_ld(rCV, us.jm, sv); tq[ts][4] = buf_ld(rCV, us.jc, sv); tq[ts][5] = buf_ld(rCV, us.jp, sv);
        }
        double p[5];
        {
            // products; rows/columns outside the image contribute 0: zeroing fx and fy zeroes all five
            const bool keep = row_in && col_in;
            double fx = f2d_x(pa0, pa1, ca0, ca1) + f2d_x(pb0, pb1, cb0, cb1);
            double fy = f2d_y(pa0, pa1, ca0, ca1) + f2d_y(pb0, pb1, cb0, cb1);
            const double ft = f2d_neg(pa0, pa1, ca0, ca1) + f2d_pos(pb0, pb1, cb0, cb1);
            fx = keep ? fx : 0.0;
            fy = keep ? fy : 0.0;
            p[0] = fx * fx; p[1] = fy * fy; p[2] = fx * fy; p[3] = fx * ft; p[4] = fy * ft;
        }
        LK3_STAMP(t1);
        if constexpr (!DPP_ONLY) {
#pragma unroll
            for (int f = 0; f < 5; ++f) {
                const double q = p[f] + dpp_swap1(p[f]);
                s_p[bsel][f][PP + t] = p[f];
                s_q[bsel][f][PQ + (t >> 1)] = q;                // both lanes of a pair store the same value
            }
            LK3_STAMP(t2);
            __syncthreads();
            LK3_STAMP(t3);
        }

        double S[5];
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            double h = 0.0;
            if constexpr (DPP_ONLY) {
                // 3-tap window inside one wave: both neighbours by DPP wave shifts, no LDS, no barrier; the nine terms in
                // get_Sum9's own order
                S[f] = sum9_step(win9[f], dpp_wave_shr1(p[f]), p[f], dpp_wave_shl1(p[f]));
            } else {
                const double *q = &s_q[bsel][f][PQ + jb];
                h = s_p[bsel][f][PP + sidx];
#pragma unroll
                for (int j = 0; j < R; ++j) h += q[j];
            }
            if constexpr (DPP_ONLY) {
            } else if constexpr (WIN == 3) {
                // (multi-wave blocks, never launched for 3 x 3) three rows summed directly, (row i-2 + row i-1) + row i
                S[f] = (ring[(K + 1) % 3][f] + ring[(K + 2) % 3][f]) + h;
                ring[K][f] = h;
            } else {
                // vertical van Herk / Gil-Werman step at static chunk position K
                pre[f] = (K == 0) ? h : pre[f] + h;
                S[f] = (K < WIN - 1) ? ring[(K + 1) % WIN][f] + pre[f] : pre[f];
                if (K > 0) ring[K][f] = h;                      // slot 0 is never read back (see below)
            }
        }
        if (WIN > 3 && K == WIN - 1) {
#pragma unroll
            for (int q = WIN - 2; q >= 1; --q)                  // suffix sums; suffix[0] (= chunk total) is never used
#pragma unroll
                for (int f = 0; f < 5; ++f) ring[q][f] += ring[q + 1][f];
        }

        LK3_STAMP(t4);
        // rotate: the only wait for the prefetch
        pa0 = ca0; pa1 = ca1; pb0 = cb0; pb1 = cb1;
        ca0 = qa0[(K + 1) % PF]; ca1 = qa1[(K + 1) % PF]; cb0 = qb0[(K + 1) % PF]; cb1 = qb1[(K + 1) % PF];
        LK3_STAMP(t5);

        if (out_row && is_out) {
            double u = 0.0, v = 0.0;
            if (y >= 1 && y <= H - 2 && x >= 1 && x <= W - 2) {
                const bool clipped = (y - R < 0) || (y + R > H - 1) || (x - R < 0) || (x + R > W - 1);
                if (clipped) {  // SerialPyr.cpp:94-95: partial window -> m(y,x) * WIN^2
                    double pc[5];
                    lk_products_at(a.i1.at(pair), a.i2.at(pair), pitch, H, W, y, x, pc);
#pragma unroll
                    for (int f = 0; f < 5; ++f) S[f] = pc[f] * (double)(WIN * WIN);
                }
                const double det = S[0] * S[1] - S[2] * S[2];
                const double un = S[2] * S[4] - S[1] * S[3];
                const double vn = S[3] * S[2] - S[0] * S[4];
                const bool nz = det != 0.0;                    // cv::divide: x/0 -> 0
                double qu, qv;
                lk_div2(un, vn, det, qu, qv);
                u = nz ? qu : 0.0;
                v = nz ? qv : 0.0;
            }
            if (HAS_UP) {
                if ((y & 1) == 0 && y > y0) {                   // wave-uniform: rotate in the row fetched last time
                    const double nu = hu[0], nv = hv[0];
                    hu[0] = hu[1]; hu[1] = hu[2]; hu[2] = nu;
                    hv[0] = hv[1]; hv[1] = hv[2]; hv[2] = nv;
                }
                if (y & 1) {
                    u += ((hu[1] + hu[2]) * 4) * (1.0 / 64);
                    v += ((hv[1] + hv[2]) * 4) * (1.0 / 64);
                    if constexpr (TAP3) {                       // u taps came with the previous (even) row, v taps with this one
                        hu[0] = pyrup_h1v(tq3[(K + 2) % 3][0], tq3[(K + 2) % 3][1], tq3[(K + 2) % 3][2], us);
                        hv[0] = pyrup_h1v(tq3[K % 3][0], tq3[K % 3][1], tq3[K % 3][2], us);
                    } else {
                        hu[0] = pyrup_h1v(tq[K % PF][0], tq[K % PF][1], tq[K % PF][2], us);   // next coarse row, used from row y+1 on
                        hv[0] = pyrup_h1v(tq[K % PF][3], tq[K % PF][4], tq[K % PF][5], us);
                    }
                } else {
                    u += (hu[0] + hu[1] * 6 + hu[2]) * (1.0 / 64);
                    v += (hv[0] + hv[1] * 6 + hv[2]) * (1.0 / 64);
                }
            }
            buf_st(rU, vxo, (y - y0) * upb, u);
            buf_st(rV, vxo, (y - y0) * vpb, v);
        }
    };

    // The van Herk chunks are anchored to ABSOLUTE image rows (chunk position of image row r = r mod WIN), not to
    // the strip: the order in which the WIN rows of a window are added is then a property of the output row alone,
    // so results do not depend on the strip height -- which the launcher picks from the batch size and the CU
    // count -- and a sharded stream reproduces the unsharded one bit for bit.  WIN = 3 sums its
    // three rows directly (no chunks; its queue slots are tied to K, so it keeps K = marched row mod 3).
    const int kc0 = WIN == 3 ? 0 : ((yy0 % WIN) + WIN) % WIN;   // chunk position of marched row 0
    for (int base = 0; base < nmarch + kc0; base += WIN) {
        [&]<int... Ks>(std::integer_sequence<int, Ks...>) __attribute__((always_inline)) {
            ((base + Ks >= kc0 && base + Ks < nmarch + kc0 ? (row(std::integral_constant<int, Ks>{}, base + Ks - kc0), 0) : 0), ...);
        }(std::make_integer_sequence<int, WIN>{});
    }
}

}  // namespace lk
