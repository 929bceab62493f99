// lk_march3.h -- single-level dense LK, row-marching kernel, one column per thread, lean body.
//
// Same stage semantics as lk_march_kernel (lk_kernels.h).  What bounds this kernel is fp64 VALU /
// LDS instruction issue, so the body is organised for instruction count and for <= 256 VGPRs
// (2 waves per SIMD):
//   * thread t owns column x = x_base + t; per row: one 8-byte load per image + one for the left
//     derivative tap (reflect-101 folded into the index), previous row carried in registers, next
//     row prefetched;
//   * horizontal WIN-tap sums: lanes 2j,2j+1 form the pair sum q_j = p(2j)+p(2j+1) with one DPP
//     quad-permute (no LDS round trip); every lane stores p, even lanes store q; a window is then
//     R pair sums + one single = R+1 LDS reads and R adds per field instead of WIN and WIN-1;
//   * ONE barrier per marched row (double-buffered LDS rows);
//   * vertical WIN-tap sums: van Herk / Gil-Werman prefix + suffix ring in REGISTERS, statically
//     indexed (dispatch on the wave-uniform chunk position); additions only -- no running
//     subtract, so flat regions stay exactly 0 like the reference;
//   * "+= pyrUp(coarse flow)": the horizontally filtered coarse rows are carried in registers and
//     refreshed every second fine row.
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

// Horizontal cv::pyrUp tap of one coarse row for ONE fine column (branch-free form of pyrup_h()):
//   even x: (A + B*mB) + C2   interior A=c[j-1] mB=6 C2=c[j+1] | left A=0 mB=6 C2=c[1]*2
//                             | right A=c[W-2] mB=7 C2=0 | single A=0 mB=8 C2=0
//   odd x : (B + Co)*4        interior/left Co=c[j+1] | right/single Co=B  ((B+B)*4 == B*8 exactly)
struct UpSel1 {
    int jm, jc, jp;
    double mB;
    bool odd, keepA, dup;
    int cmode;  // C2 = 0 (0), c[jp] (1), c[jp]*2 (2)
};
__device__ __forceinline__ UpSel1 make_upsel1(int x, int Wc)
{
    UpSel1 s;
    int j = x >> 1;
    j = j < 0 ? 0 : (j > Wc - 1 ? Wc - 1 : j);
    s.jc = j;
    s.jm = j > 0 ? j - 1 : 0;
    s.jp = j < Wc - 1 ? j + 1 : Wc - 1;
    const bool single = Wc == 1, left = j == 0, right = j == Wc - 1;
    s.odd = (x & 1) != 0;
    s.mB = single ? 8.0 : (right ? 7.0 : 6.0);
    s.keepA = !(single || left);
    s.cmode = (single || right) ? 0 : (left ? 2 : 1);
    s.dup = single || right;
    return s;
}
__device__ __forceinline__ double pyrup_h1v(double A, double B, double C, const UpSel1 &s);
__device__ __forceinline__ double pyrup_h1(const double *c, const UpSel1 &s)
{
    return pyrup_h1v(c[s.jm], c[s.jc], c[s.jp], s);
}
__device__ __forceinline__ double pyrup_h1v(double A, double B, double C, const UpSel1 &s)
{
    // branch-free: both parities are evaluated, the lane's own is selected (odd/even lanes alternate,
    // so a branch would execute both sides anyway)
    const double A1 = s.keepA ? A : 0.0;
    const double C2 = s.cmode == 2 ? C * 2 : (s.cmode == 1 ? C : 0.0);
    const double ev = (A1 + B * s.mB) + C2;
    const double od = (B + (s.dup ? B : C)) * 4;
    return s.odd ? od : ev;
}

template <int WIN, int TX, bool HAS_UP>
__global__ __launch_bounds__(TX, (WIN <= 15 ? 2 : 1)) void lk_march3_kernel(LKArgs a)
{
    constexpr int R = WIN / 2;
    constexpr int XO = TX - 2 * R;            // output columns per block
    constexpr int PP = 16, PQ = 8;            // LDS pads (only non-output threads read them)
    __shared__ double s_p[2][5][TX + 2 * PP];
    __shared__ double s_q[2][5][TX / 2 + 2 * PQ];

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
    const double *I1c = a.i1.at(pair) + xc, *I2c = a.i2.at(pair) + xc;   // own column
    const double *I1l = a.i1.at(pair) + xl, *I2l = a.i2.at(pair) + xl;   // left derivative tap
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

    // previous row (pa*, pb*) and prefetched current row (ca*, cb*): [0] = left tap, [1] = own
    int ro = r101(yy0 - 1, H) * pitch;
    double pa0 = I1l[ro], pa1 = I1c[ro], pb0 = I2l[ro], pb1 = I2c[ro];
    ro = r101(yy0, H) * pitch;
    double ca0 = I1l[ro], ca1 = I1c[ro], cb0 = I2l[ro], cb1 = I2c[ro];

    [[maybe_unused]] UpSel1 us{};
    [[maybe_unused]] double hu[3] = {0, 0, 0}, hv[3] = {0, 0, 0};   // carried coarse rows s(i-1),s(i),s(i+1)
    [[maybe_unused]] const int Hc = H >> 1;
    [[maybe_unused]] const double *CU = nullptr, *CV = nullptr;
    [[maybe_unused]] const int cpu = (int)a.cu.pitch, cpv = (int)a.cv.pitch;
    auto up_srow = [&](int k) { return k < 0 ? min(1, Hc - 1) : (k > Hc - 1 ? Hc - 1 : k); };
    if (HAS_UP) {
        us = make_upsel1(xc, W >> 1);
        CU = a.cu.at(pair);
        CV = a.cv.at(pair);
        const int i0 = y0 >> 1;                                 // slots of the first output row
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int sr = up_srow(i0 - 1 + s);
            hu[s] = pyrup_h1(CU + sr * cpu, us);
            hv[s] = pyrup_h1(CV + sr * cpv, us);
        }
    }

    // One marched row at chunk position K (compile-time): the chunk loop below is unrolled WIN
    // times so that every ring access is a plain register (no dispatch, no copies).
#ifdef LK3_STAMPS
    unsigned long long seg[6] = {0, 0, 0, 0, 0, 0};
#define LK3_STAMP(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#else
#define LK3_STAMP(var)
#endif
    auto row = [&](auto kc, int i) __attribute__((always_inline)) {
        constexpr int K = decltype(kc)::value;
        LK3_STAMP(t0);
        const int bsel = i & 1;
        const int yy = yy0 + i;
        const bool row_in = yy >= 0 && yy < H;
        // prefetch the next image row (clamped, unconditional); consumed by the rotate below, after
        // the window sums, so that its latency hides behind this row's work
        // apron rows carry the reflected row (at yy = 0 the "previous" row is row 1 = r101(-1));
        // one reflection suffices unless the image is shorter than the apron
        int rr = yy + 1;
        rr = rr < 0 ? -rr : rr;
        rr = rr >= H ? 2 * H - 2 - rr : rr;
        if (H <= R + 1) rr = r101(yy + 1, H);
        const int rn = rr * pitch;
        const double na0 = I1l[rn], na1 = I1c[rn], nb0 = I2l[rn], nb1 = I2c[rn];
        const int y = yy - R;                                   // output row of this iteration
        const bool out_row = i >= 2 * R;
        // the NEXT coarse row of the "+= pyrUp" term is turned into the new carried slot in the epilogue of
        // odd output rows (slot 0 is dead there).  Its three taps per plane are fetched at the top of EVERY
        // row, unconditionally: with a conditional fetch the compiler's s_waitcnt before the products
        // (which only need the row prefetched one iteration ago) counts the minimum over both paths and
        // ends up waiting for the image loads issued just above on every odd row.
        [[maybe_unused]] double uA = 0, uB = 0, uC = 0, vA = 0, vB = 0, vC = 0;
        if (HAS_UP) {
            const int sr = up_srow(((y + 1) >> 1) + 1);
            const double *ru = CU + sr * cpu, *rv = CV + sr * cpv;
            uA = ru[us.jm]; uB = ru[us.jc]; uC = ru[us.jp];
            vA = rv[us.jm]; vB = rv[us.jc]; vC = rv[us.jp];
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
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            const double q = p[f] + dpp_swap1(p[f]);
            s_p[bsel][f][PP + t] = p[f];
            s_q[bsel][f][PQ + (t >> 1)] = q;                    // both lanes of a pair store the same value
        }
        LK3_STAMP(t2);
        __syncthreads();
        LK3_STAMP(t3);

        double S[5];
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            const double *q = &s_q[bsel][f][PQ + jb];
            double h = s_p[bsel][f][PP + sidx];
#pragma unroll
            for (int j = 0; j < R; ++j) h += q[j];
            // vertical van Herk / Gil-Werman step at static chunk position K
            pre[f] = (K == 0) ? h : pre[f] + h;
            S[f] = (K < WIN - 1) ? ring[(K + 1) % WIN][f] + pre[f] : pre[f];
            if (K > 0) ring[K][f] = h;                          // slot 0 is never read back (see below)
        }
        if (K == WIN - 1) {
#pragma unroll
            for (int q = WIN - 2; q >= 1; --q)                  // suffix sums; suffix[0] (= chunk total) is never used
#pragma unroll
                for (int f = 0; f < 5; ++f) ring[q][f] += ring[q + 1][f];
        }

        LK3_STAMP(t4);
        // rotate: the only wait for the prefetch
        pa0 = ca0; pa1 = ca1; pb0 = cb0; pb1 = cb1;
        ca0 = na0; ca1 = na1; cb0 = nb0; cb1 = nb1;
#ifdef LK3_STAMPS
        asm volatile("" :: "v"(pa0), "v"(ca0), "v"(ca1), "v"(cb0), "v"(cb1));
#endif
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
                const double qu = un / det, qv = vn / det;
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
                    hu[0] = pyrup_h1v(uA, uB, uC, us);          // next coarse row, used from row y+1 on
                    hv[0] = pyrup_h1v(vA, vB, vC, us);
                } else {
                    u += (hu[0] + hu[1] * 6 + hu[2]) * (1.0 / 64);
                    v += (hv[0] + hv[1] * 6 + hv[2]) * (1.0 / 64);
                }
            }
            a.u.at(pair)[y * (int)a.u.pitch + x] = u;
            a.v.at(pair)[y * (int)a.v.pitch + x] = v;
        }
#ifdef LK3_STAMPS
        LK3_STAMP(t6);
        seg[0] += t1 - t0; seg[1] += t2 - t1; seg[2] += t3 - t2; seg[3] += t4 - t3; seg[4] += t5 - t4; seg[5] += t6 - t5;
#endif
    };

    for (int base = 0; base < nmarch; base += WIN) {
        [&]<int... Ks>(std::integer_sequence<int, Ks...>) __attribute__((always_inline)) {
            ((base + Ks < nmarch ? (row(std::integral_constant<int, Ks>{}, base + Ks), 0) : 0), ...);
        }(std::make_integer_sequence<int, WIN>{});
    }
#ifdef LK3_STAMPS
    // diagnostic build only: per-segment cycle totals of wave 0 of every block -> debug buffer (a.dbg)
    if (a.dbg && (threadIdx.x & 63) == 0) {
        const unsigned wid = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (TX / 64) + (threadIdx.x >> 6);
        for (int k = 0; k < 6; ++k) a.dbg[wid * 8 + k] = seg[k];
        a.dbg[wid * 8 + 6] = nmarch;
    }
#endif
}

}  // namespace lk
