// lk_device.h -- per-pixel device primitives shared by every kernel (gfx950, fp64).
//
// Each function restates one row of SURVEY.md 8(a); expression ORDER mirrors the OpenCV 2.4
// templates the reference calls, and the translation unit is compiled with -ffp-contract=off so
// that no product is fused into an FMA (the reference rounds every product).
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>

namespace lk {

// A batch of strided fp64 planes: element (p, y, x) lives at base[p*plane + y*pitch + x].
struct Planes {
    double *base;
    long long plane;  // elements between planes
    long long pitch;  // elements between rows
    __device__ __forceinline__ double *at(int p) const { return base + (long long)p * plane; }
};
struct CPlanes {
    const double *base;
    long long plane;
    long long pitch;
    __device__ __forceinline__ const double *at(int p) const { return base + (long long)p * plane; }
};

// cv::borderInterpolate(p, len, BORDER_REFLECT_101)
__device__ __forceinline__ int r101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// The three 2x2 filter2D responses of ONE image (anchor (1,1)): p00=I(y-1,x-1) p01=I(y-1,x)
// p10=I(y,x-1) p11=I(y,x).  ParallelVideoPyr.cpp:44-84; accumulation order of OpenCV 2.4
// Filter2D (s = 0; s += k*tap in row-major tap order).
__device__ __forceinline__ double f2d_x(double p00, double p01, double p10, double p11)
{
    return ((-p00 + p01) - p10) + p11;  // [[-1, 1], [-1, 1]]
}
__device__ __forceinline__ double f2d_y(double p00, double p01, double p10, double p11)
{
    return ((-p00 - p01) + p10) + p11;  // [[-1,-1], [ 1, 1]]
}
__device__ __forceinline__ double f2d_neg(double p00, double p01, double p10, double p11)
{
    return ((-p00 - p01) - p10) - p11;  // all -1 (image 1 of get_ft)
}
__device__ __forceinline__ double f2d_pos(double p00, double p01, double p10, double p11)
{
    return ((p00 + p01) + p10) + p11;  // all +1 (image 2 of get_ft)
}

// fx, fy, ft and the five products fx2, fy2, fxfy, fxft, fyft (ParallelVideoPyr.cpp:378-405).
// a** = image 1 taps, b** = image 2 taps.
__device__ __forceinline__ void lk_products(double a00, double a01, double a10, double a11,
                                            double b00, double b01, double b10, double b11,
                                            double p[5])
{
    const double fx = f2d_x(a00, a01, a10, a11) + f2d_x(b00, b01, b10, b11);
    const double fy = f2d_y(a00, a01, a10, a11) + f2d_y(b00, b01, b10, b11);
    const double ft = f2d_neg(a00, a01, a10, a11) + f2d_pos(b00, b01, b10, b11);
    p[0] = fx * fx;
    p[1] = fy * fy;
    p[2] = fx * fy;
    p[3] = fx * ft;
    p[4] = fy * ft;
}

// Products at (y,x) straight from global memory (used for the clipped-window rule and by the
// naive kernel).  Reflect-101 on the -1 row/column.
__device__ __forceinline__ void lk_products_at(const double *I1, const double *I2, long long pitch,
                                               int H, int W, int y, int x, double p[5])
{
    const int ym = r101(y - 1, H), xm = r101(x - 1, W);
    const double *r0a = I1 + (long long)ym * pitch, *r1a = I1 + (long long)y * pitch;
    const double *r0b = I2 + (long long)ym * pitch, *r1b = I2 + (long long)y * pitch;
    lk_products(r0a[xm], r0a[x], r1a[xm], r1a[x], r0b[xm], r0b[x], r1b[xm], r1b[x], p);
}

// x1 / y and x2 / y, both correctly rounded (IEEE-754 binary64, every input class), sharing the reciprocal iteration.
// hipcc expands an fp64 division into  d = div_scale(den) ; r = rcp(d) + two Newton steps (4 FMA) ; n = div_scale(num) ;
// q = n r ; rem = fma(-d, q, n) ; div_fmas ; div_fixup  -- eleven instructions, of which the first six depend on the numerator
// only through div_scale's choice of a scale for extreme exponent differences.  The two divisions of the LK solve have one
// denominator (det): when the scaled denominators of both come out bit-equal in every lane of the wave (always, outside
// exponent differences of several hundred binades) the reciprocal is refined once and each quotient takes its own five
// closing instructions -- the same instructions on the same inputs as the two full expansions, hence the same bits; otherwise
// (wave-uniform branch) the two full divisions run.  17 instructions and one v_rcp_f64 per pixel instead of 22 and two.
__device__ __forceinline__ void lk_div2(double x1, double x2, double y, double &q1, double &q2)
{
    bool f1, f2, fd;
    const double d1 = __builtin_amdgcn_div_scale(x1, y, false, &fd);
    const double d2 = __builtin_amdgcn_div_scale(x2, y, false, &fd);
    if (__builtin_amdgcn_ballot_w64(__double_as_longlong(d1) != __double_as_longlong(d2)) == 0) {
        const double r0 = __builtin_amdgcn_rcp(d1);
        const double e0 = __builtin_fma(-d1, r0, 1.0);
        const double r1 = __builtin_fma(r0, e0, r0);
        const double e1 = __builtin_fma(-d1, r1, 1.0);
        const double r2 = __builtin_fma(r1, e1, r1);
        const double n1 = __builtin_amdgcn_div_scale(x1, y, true, &f1);
        const double n2 = __builtin_amdgcn_div_scale(x2, y, true, &f2);
        const double m1 = n1 * r2, m2 = n2 * r2;
        const double rem1 = __builtin_fma(-d1, m1, n1), rem2 = __builtin_fma(-d1, m2, n2);
        q1 = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(rem1, r2, m1, f1), y, x1);
        q2 = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(rem2, r2, m2, f2), y, x2);
    } else {
        q1 = x1 / y;
        q2 = x2 / y;
    }
}

// 2x2 solve + OpenCV-2.4 divide (den == 0 -> 0).  ParallelVideoPyr.cpp:470-472, 510-520.
// s = {Sxx, Syy, Sxy, Sxt, Syt}
__device__ __forceinline__ void lk_solve(const double s[5], double &u, double &v)
{
    const double det = s[0] * s[1] - s[2] * s[2];
    const double un = s[2] * s[4] - s[1] * s[3];
    const double vn = s[3] * s[2] - s[0] * s[4];
    double qu, qv;
    lk_div2(un, vn, det, qu, qv);
    u = det != 0.0 ? qu : 0.0;
    v = det != 0.0 ? qv : 0.0;
}
// the same with two plain divisions (the compiler's expansion): lkflow_solve_f64_dev's cross-check mode
__device__ __forceinline__ void lk_solve_plain(const double s[5], double &u, double &v)
{
    const double det = s[0] * s[1] - s[2] * s[2];
    const double un = s[2] * s[4] - s[1] * s[3];
    const double vn = s[3] * s[2] - s[0] * s[4];
    u = det != 0.0 ? un / det : 0.0;
    v = det != 0.0 ? vn / det : 0.0;
}

// The reference's 3 x 3 window sum, get_Sum9 (ParallelVideoPyr.cpp:114-160, loop form SerialPyr.cpp:80-96): val = 0, then
// nine sequential additions in row-major order, ((((((((t0 + t1) + t2) + m0) + m1) + m2) + b0) + b1) + b2).  A row-marching
// kernel meets the rows of a window one at a time, so per field and column it carries the partial of the window whose TOP
// row was the previous marched row (a) and of the window whose top and middle rows were the two previous rows (b); the
// marched row (l, c, r) completes the second, extends the first and opens a new one: 8 additions per field against 4 for
// the separable order of rounds 1-3, and in return the 3 x 3 path has the reference's bits wherever a window holds nothing
// but rounding residue (straight edges, singular structure tensors -- the reference's own edge-map inputs).  The result
// does not depend on where a strip starts.  (0.0 + t0 is t0 up to the sign of a zero.)
struct Sum9 { double a, b; };
__device__ __forceinline__ double sum9_step(Sum9 &s, double l, double c, double r)
{
    const double S = ((s.b + l) + c) + r;
    s.b = ((s.a + l) + c) + r;
    s.a = (l + c) + r;
    return S;
}

// Horizontal pass of cv::pyrUp at fine column x for one coarse row c[0..Wc) (OpenCV 2.4
// pyramids.cpp pyrUp_; asymmetric borders exactly as the template writes them).
__device__ __forceinline__ double pyrup_h(const double *c, int Wc, int x)
{
    const int j = x >> 1;
    if (Wc == 1) return c[0] * 8;
    if ((x & 1) == 0) {
        if (j == 0) return c[0] * 6 + c[1] * 2;
        if (j == Wc - 1) return c[Wc - 2] + c[Wc - 1] * 7;
        return c[j - 1] + c[j] * 6 + c[j + 1];
    }
    if (j == Wc - 1) return c[Wc - 1] * 8;
    return (c[j] + c[j + 1]) * 4;
}

// cv::pyrUp value at fine pixel (y,x) of a coarse Hc x Wc plane (ParallelVideoPyr.cpp:646-647).
__device__ __forceinline__ double pyrup_at(const double *c, long long pitch, int Hc, int Wc, int y,
                                           int x)
{
    const int i = y >> 1;
    const int s1 = r101(2 * i, 2 * Hc) >> 1;
    const int s2 = r101(2 * (i + 1), 2 * Hc) >> 1;
    const double r1 = pyrup_h(c + (long long)s1 * pitch, Wc, x);
    const double r2 = pyrup_h(c + (long long)s2 * pitch, Wc, x);
    if (y & 1) return ((r1 + r2) * 4) * (1.0 / 64);
    const int s0 = r101(2 * (i - 1), 2 * Hc) >> 1;
    const double r0 = pyrup_h(c + (long long)s0 * pitch, Wc, x);
    return (r0 + r1 * 6 + r2) * (1.0 / 64);
}

__device__ __forceinline__ int sat_short(int v)
{
    return v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
}

// cvRound(t) as OpenCV 2.4 compiles it on x86-64 (cvtsd2si: round-half-even; NaN / out of range ->
// 0x80000000), for the FLOAT argument remap passes (map coordinate * 32): float -> double is exact, so
// rounding the float half-to-even gives the same integer as rounding its double image, at 32-bit cost.
__device__ __forceinline__ int cv_round_f(float t)
{
    if (!(fabsf(t) < 2147483648.0f)) return INT_MIN;   // NaN / out of range -> 0x80000000
    return __float2int_rn(t);
}

// cv::remap INTER_LINEAR / BORDER_CONSTANT(0) sample of src at float map coordinate (mx,my)
// (OpenCV 2.4 imgwarp.cpp: 1/32-pixel fixed point, float weight table, fp64 blend, taps outside the
// image contribute 0).  When all four taps are inside, the two horizontal taps of a row are fetched
// as one 16-byte load.
__device__ __forceinline__ double remap_bilinear(const double *src, int pitch, int H, int W, float mx,
                                                    float my)
{
    const int sx = cv_round_f(mx * 32.0f);
    const int sy = cv_round_f(my * 32.0f);
    const int ix = sat_short(sx >> 5), iy = sat_short(sy >> 5);
    const float a = (float)(sx & 31) * (1.0f / 32), b = (float)(sy & 31) * (1.0f / 32);
    const float w0 = (1.0f - b) * (1.0f - a), w1 = (1.0f - b) * a, w2 = b * (1.0f - a), w3 = b * a;
    double v0, v1, v2, v3;
    if ((unsigned)ix < (unsigned)(W - 1) && (unsigned)iy < (unsigned)(H - 1)) {   // all four taps inside
        const double *r0 = src + iy * pitch + ix;
        // 8-byte aligned 16-byte loads: dword-aligned multi-dword global loads are legal on gfx9
        typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
        const d2u t0 = *reinterpret_cast<const d2u *>(r0);
        const d2u t1 = *reinterpret_cast<const d2u *>(r0 + pitch);
        v0 = t0.x; v1 = t0.y; v2 = t1.x; v3 = t1.y;
    } else {
        const bool x0 = ix >= 0 && ix < W, x1 = ix + 1 >= 0 && ix + 1 < W;
        const bool y0 = iy >= 0 && iy < H, y1 = iy + 1 >= 0 && iy + 1 < H;
        const double *r0 = src + (long long)iy * pitch + ix;
        const double *r1 = r0 + pitch;
        v0 = (y0 && x0) ? r0[0] : 0.0;
        v1 = (y0 && x1) ? r0[1] : 0.0;
        v2 = (y1 && x0) ? r1[0] : 0.0;
        v3 = (y1 && x1) ? r1[1] : 0.0;
    }
    return v0 * (double)w0 + v1 * (double)w1 + v2 * (double)w2 + v3 * (double)w3;
}

// The same sample for TWO independent (image, coordinate) pairs with every gather issued UNCONDITIONALLY and up front: the
// taps are fetched from addresses clamped into the image (H, W >= 2) and the ones that fall outside are zeroed afterwards
// (BORDER_CONSTANT 0) -- no load sits under a condition, so hipcc's vmcnt counting stays exact and the four 16-byte
// gathers fly together (remap_bilinear's conditional loads make it drain vmcnt(0) in front of each one: four serial memory
// round trips per marched row in a kernel that has only two waves per SIMD to hide them).  Same fixed-point arithmetic,
// same blend expression: bit-identical to remap_bilinear().
// (round 4: the record carries what the blend needs -- the two 5-bit fractions as bytes 0 and 1 of one register, which
// v_cvt_f32_ubyte0 / 1 convert without an unpack, and the distance of the real tap block from the clamped one -- instead of
// the raw fixed-point coordinates, from which the blend re-derived both: 12 vector instructions per sample fewer.)
struct RemapTaps { int fr; int off; int dxy; };   // fractions (sx & 31) | (sy & 31) << 8; element offset of the clamped block; (dx & 0xFFFF) | dy << 16
template <bool M24 = false>
__device__ __forceinline__ RemapTaps remap_taps(int pitch, int H, int W, float mx, float my)
{
    RemapTaps t;
    const int sx = cv_round_f(mx * 32.0f), sy = cv_round_f(my * 32.0f);
    const int ix = sat_short(sx >> 5), iy = sat_short(sy >> 5);       // cv::remap keeps the integer parts as shorts
    const int ixc = min(max(ix, 0), W - 2), iyc = min(max(iy, 0), H - 2);
    // M24 (host-checked by the launcher: pitch and H below 2^24): full-rate 24-bit multiply instead of v_mul_lo_u32
    t.off = M24 ? (int)__umul24((unsigned)iyc, (unsigned)pitch) + ixc : iyc * pitch + ixc;
    t.fr = (sx & 31) | ((sy & 31) << 8);
    // both differences lie in [-32768, 32767] (ix < 0 -> ixc = 0; ix > W - 2 -> ixc = W - 2 >= 0): they pack into halves
    t.dxy = ((ix - ixc) & 0xFFFF) | ((iy - iyc) << 16);
    return t;
}
typedef double lk_d2u __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ double remap_blend(const RemapTaps &t, lk_d2u top, lk_d2u bot, int, int)
{
    const float a = (float)(t.fr & 0xFF) * (1.0f / 32), b = (float)((t.fr >> 8) & 0xFF) * (1.0f / 32);
    const float w0 = (1.0f - b) * (1.0f - a), w1 = (1.0f - b) * a, w2 = b * (1.0f - a), w3 = b * a;
    if (t.dxy != 0) {                                        // some tap of this lane leaves the image (no memory instruction inside)
        const int dx = (int)(short)(t.dxy & 0xFFFF), dy = t.dxy >> 16;
        const lk_d2u z = {0.0, 0.0};
        const lk_d2u t0 = top, t1 = bot;
        top = dy == 0 ? t0 : (dy == 1 ? t1 : z);
        bot = dy == 0 ? t1 : (dy == -1 ? t0 : z);
        const lk_d2u a0 = top, a1 = bot;
        top.x = dx == 0 ? a0.x : (dx == 1 ? a0.y : 0.0); top.y = dx == 0 ? a0.y : (dx == -1 ? a0.x : 0.0);
        bot.x = dx == 0 ? a1.x : (dx == 1 ? a1.y : 0.0); bot.y = dx == 0 ? a1.y : (dx == -1 ? a1.x : 0.0);
    }
    return top.x * (double)w0 + top.y * (double)w1 + bot.x * (double)w2 + bot.y * (double)w3;
}
__device__ __forceinline__ void remap_bilinear_x2(const double *srcA, const double *srcB, int pitch, int H, int W, float mxA,
                                                  float myA, float mxB, float myB, double &outA, double &outB)
{
    const RemapTaps ta = remap_taps(pitch, H, W, mxA, myA), tb = remap_taps(pitch, H, W, mxB, myB);
    const lk_d2u a0 = *reinterpret_cast<const lk_d2u *>(srcA + ta.off), a1 = *reinterpret_cast<const lk_d2u *>(srcA + ta.off + pitch);
    const lk_d2u b0 = *reinterpret_cast<const lk_d2u *>(srcB + tb.off), b1 = *reinterpret_cast<const lk_d2u *>(srcB + tb.off + pitch);
    outA = remap_blend(ta, a0, a1, H, W);
    outB = remap_blend(tb, b0, b1, H, W);
}

// Raw buffer access (gfx9 V#: stride 0, offset = VGPR byte offset + SGPR byte offset, range-checked
// against num_records).  The per-thread column offset is loop-invariant and the row offset is
// wave-uniform, so a row access costs NO vector ALU address arithmetic.
typedef unsigned int lk_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const double *base, long long bytes)
{
    const unsigned n = bytes > 0xFFFFFFFFll ? 0xFFFFFFFFu : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(base), 0, n, 0x00020000);
}
__device__ __forceinline__ double buf_ld(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    const lk_u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return __hiloint2double((int)v.y, (int)v.x);
}
__device__ __forceinline__ void buf_st_cached(__amdgpu_buffer_rsrc_t r, int voff, int soff, double d)   // re-read soon
{
    lk_u2 v;
    v.x = (unsigned)__double2loint(d);
    v.y = (unsigned)__double2hiint(d);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, int voff, int soff, double d)
{
    lk_u2 v;
    v.x = (unsigned)__double2loint(d);
    v.y = (unsigned)__double2hiint(d);
    // aux = 2 ("nt"): streaming store.  Every plane the pipeline writes is consumed by a LATER kernel, long
    // after 2 GB of other traffic, so keeping the lines in L2 / Infinity Cache only evicts useful ones
    // (A/B on 32 x 1080p: warp 0.595 -> 0.572 ms, whole step -1.7 %).
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 2);
}

}  // namespace lk
