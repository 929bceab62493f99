/*
 * lk_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see lk_oracle.h).
 *
 * Serial, plain C, fp64.  Follows /root/reference/SerialPyr.cpp function by function and
 * restates the OpenCV 2.4.x routines it calls (OpenCV is a third-party dependency that is NOT
 * vendored in the reference; pinned only by Makefile:8-10 "-lopencv_legacy -lopencv_gpu ..." =
 * the 2.4 series).  Operation ORDER inside every expression mirrors the OpenCV 2.4 templates so
 * that rounding matches as closely as a restatement can.
 *
 * Build with -ffp-contract=off: the reference (x86-64, SSE2, no FMA) rounds every product.
 */
#include "lk_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* cv::borderInterpolate(p, len, BORDER_REFLECT_101) */
static int r101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

/* SerialPyr.cpp:196-215 */
int lko_max_layer(int cols, int rows)
{
    int p = 1, res;
    if (cols <= 0 || rows <= 0) return 0;
    while (cols % (1 << p) == 0) ++p;
    res = p;
    p = 1;
    while (rows % (1 << p) == 0) ++p;
    return res < p ? res : p;
}

/*
 * cv::filter2D(src, dst, -1, K) for a 2x2 CV_64F kernel, default anchor -> (1,1), default border
 * REFLECT_101, delta 0.  OpenCV 2.4 filter.cpp Filter2D<double,Cast<double,double>,FilterNoVec>:
 * s0 = delta; for each non-zero kernel tap in row-major order: s0 += kf[k]*src[...].
 */
static void filter2d_2x2(const double *src, int rows, int cols, double k00, double k01, double k10,
                         double k11, double *dst)
{
    for (int y = 0; y < rows; ++y) {
        const double *r0 = src + (size_t)r101(y - 1, rows) * cols;
        const double *r1 = src + (size_t)y * cols;
        for (int x = 0; x < cols; ++x) {
            int xm = r101(x - 1, cols);
            double s = 0.0;
            s += k00 * r0[xm];
            s += k01 * r0[x];
            s += k10 * r1[xm];
            s += k11 * r1[x];
            dst[(size_t)y * cols + x] = s;
        }
    }
}

static void add_planes(const double *a, const double *b, size_t n, double *dst)
{
    for (size_t i = 0; i < n; ++i) dst[i] = a[i] + b[i];
}

/* SerialPyr.cpp:31-43 : kernel [[-1,1],[-1,1]] on both images, fx = dst1 + dst2 */
void lko_get_fx(const double *i1, const double *i2, int rows, int cols, double *fx)
{
    size_t n = (size_t)rows * cols;
    double *d1 = (double *)malloc(n * sizeof(double)), *d2 = (double *)malloc(n * sizeof(double));
    filter2d_2x2(i1, rows, cols, -1.0, 1.0, -1.0, 1.0, d1);
    filter2d_2x2(i2, rows, cols, -1.0, 1.0, -1.0, 1.0, d2);
    add_planes(d1, d2, n, fx);
    free(d1); free(d2);
}

/* SerialPyr.cpp:45-57 : kernel [[-1,-1],[1,1]] */
void lko_get_fy(const double *i1, const double *i2, int rows, int cols, double *fy)
{
    size_t n = (size_t)rows * cols;
    double *d1 = (double *)malloc(n * sizeof(double)), *d2 = (double *)malloc(n * sizeof(double));
    filter2d_2x2(i1, rows, cols, -1.0, -1.0, 1.0, 1.0, d1);
    filter2d_2x2(i2, rows, cols, -1.0, -1.0, 1.0, 1.0, d2);
    add_planes(d1, d2, n, fy);
    free(d1); free(d2);
}

/* SerialPyr.cpp:59-71 : all -1 on image 1, all +1 on image 2 */
void lko_get_ft(const double *i1, const double *i2, int rows, int cols, double *ft)
{
    size_t n = (size_t)rows * cols;
    double *d1 = (double *)malloc(n * sizeof(double)), *d2 = (double *)malloc(n * sizeof(double));
    filter2d_2x2(i1, rows, cols, -1.0, -1.0, -1.0, -1.0, d1);
    filter2d_2x2(i2, rows, cols, 1.0, 1.0, 1.0, 1.0, d2);
    add_planes(d1, d2, n, ft);
    free(d1); free(d2);
}

/*
 * SerialPyr.cpp:80-96 get_Sum9 with the 3x3 loops widened to win x win ("minimal edit":
 * -1..1 -> -r..r, 9 -> win*win).  Accumulation order is the reference's: row-major, sequential.
 */
static double sum_win_at(const double *m, int rows, int cols, int y, int x, int r)
{
    if (x < 0 || x >= cols) return 0;
    if (y < 0 || y >= rows) return 0;
    double val = 0.0;
    int tmp = 0;
    for (int i = -r; i <= r; ++i) {
        for (int j = -r; j <= r; ++j) {
            int yy = y + i, xx = x + j;
            if (xx >= 0 && xx < cols && yy >= 0 && yy < rows) {
                ++tmp;
                val += m[(size_t)yy * cols + xx];
            }
        }
    }
    int full = (2 * r + 1) * (2 * r + 1);
    if (tmp == full) return val;
    return m[(size_t)y * cols + x] * full;
}

/* SerialPyr.cpp:98-106 get_Sum9_Mat: zeros, then rows 1..H-2, cols 1..W-2 */
void lko_sum_win(const double *m, int rows, int cols, int win, double *dst)
{
    int r = win / 2;
    memset(dst, 0, (size_t)rows * cols * sizeof(double));
    for (int i = 1; i < rows - 1; ++i)
        for (int j = 1; j < cols - 1; ++j)
            dst[(size_t)i * cols + j] = sum_win_at(m, rows, cols, i, j, r);
}

/* SerialPyr.cpp:121-144 */
void lko_lucas_kanade(const double *i1, const double *i2, int rows, int cols, int win, double *u,
                      double *v)
{
    size_t n = (size_t)rows * cols;
    double *buf = (double *)malloc(13 * n * sizeof(double));
    double *fx = buf, *fy = buf + n, *ft = buf + 2 * n;
    double *fx2 = buf + 3 * n, *fy2 = buf + 4 * n, *fxfy = buf + 5 * n, *fxft = buf + 6 * n,
           *fyft = buf + 7 * n;
    double *sfx2 = buf + 8 * n, *sfy2 = buf + 9 * n, *sfxft = buf + 10 * n, *sfxfy = buf + 11 * n,
           *sfyft = buf + 12 * n;

    lko_get_fx(i1, i2, rows, cols, fx);
    lko_get_fy(i1, i2, rows, cols, fy);
    lko_get_ft(i1, i2, rows, cols, ft);
    for (size_t i = 0; i < n; ++i) {           /* :127-131 Mat::mul */
        fx2[i] = fx[i] * fx[i];
        fy2[i] = fy[i] * fy[i];
        fxfy[i] = fx[i] * fy[i];
        fxft[i] = fx[i] * ft[i];
        fyft[i] = fy[i] * ft[i];
    }
    lko_sum_win(fx2, rows, cols, win, sfx2);   /* :133-137 */
    lko_sum_win(fy2, rows, cols, win, sfy2);
    lko_sum_win(fxft, rows, cols, win, sfxft);
    lko_sum_win(fxfy, rows, cols, win, sfxfy);
    lko_sum_win(fyft, rows, cols, win, sfyft);
    for (size_t i = 0; i < n; ++i) {           /* :139-143 */
        double tmp = sfx2[i] * sfy2[i] - sfxfy[i] * sfxfy[i];
        double un = sfxfy[i] * sfyft[i] - sfy2[i] * sfxft[i];
        double vn = sfxft[i] * sfxfy[i] - sfx2[i] * sfyft[i];
        /* cv::divide (OpenCV 2.4 arithm.cpp div_): denominator 0 -> 0 */
        u[i] = tmp != 0 ? un / tmp : 0.0;
        v[i] = tmp != 0 ? vn / tmp : 0.0;
    }
    free(buf);
}

/*
 * cv::pyrDown, OpenCV 2.4 pyramids.cpp pyrDown_<FltCast<double,8>,NoVec>, BORDER_DEFAULT
 * (REFLECT_101): horizontal [1 4 6 4 1] with decimation into a row buffer, then vertical,
 * then * 1/256.  Expression order as in the template:
 *   row[x] = s[2x]*6 + (s[2x-1] + s[2x+1])*4 + s[2x-2] + s[2x+2]
 *   dst[x] = (row2*6 + (row1 + row3)*4 + row0 + row4) * (1/256)
 */
void lko_pyr_down(const double *src, int rows, int cols, double *dst)
{
    int drows = (rows + 1) / 2, dcols = (cols + 1) / 2;
    double *hbuf = (double *)malloc((size_t)rows * dcols * sizeof(double));
    for (int y = 0; y < rows; ++y) {
        const double *s = src + (size_t)y * cols;
        double *row = hbuf + (size_t)y * dcols;
        for (int x = 0; x < dcols; ++x) {
            double c = s[r101(2 * x, cols)];
            double l1 = s[r101(2 * x - 1, cols)], r1 = s[r101(2 * x + 1, cols)];
            double l2 = s[r101(2 * x - 2, cols)], r2 = s[r101(2 * x + 2, cols)];
            row[x] = c * 6 + (l1 + r1) * 4 + l2 + r2;
        }
    }
    for (int y = 0; y < drows; ++y) {
        const double *row0 = hbuf + (size_t)r101(2 * y - 2, rows) * dcols;
        const double *row1 = hbuf + (size_t)r101(2 * y - 1, rows) * dcols;
        const double *row2 = hbuf + (size_t)r101(2 * y, rows) * dcols;
        const double *row3 = hbuf + (size_t)r101(2 * y + 1, rows) * dcols;
        const double *row4 = hbuf + (size_t)r101(2 * y + 2, rows) * dcols;
        double *d = dst + (size_t)y * dcols;
        for (int x = 0; x < dcols; ++x)
            d[x] = (row2[x] * 6 + (row1[x] + row3[x]) * 4 + row0[x] + row4[x]) * (1.0 / 256);
    }
    free(hbuf);
}

/*
 * cv::pyrUp, OpenCV 2.4 pyramids.cpp pyrUp_<FltCast<double,6>,NoVec>, dsize = 2*ssize.
 * Horizontal pass per source row (asymmetric borders, exactly as the template writes them):
 *   x == 0     : even = s[0]*6 + s[1]*2        odd = (s[0] + s[1])*4
 *   x == W-1   : even = s[W-2] + s[W-1]*7      odd = s[W-1]*8
 *   otherwise  : even = s[x-1] + s[x]*6 + s[x+1]   odd = (s[x] + s[x+1])*4
 *   W == 1     : even = odd = s[0]*8
 * Vertical pass with source rows _sy = borderInterpolate(sy*2, 2H, REFLECT_101)/2:
 *   dst(2y)   = (row0 + row1*6 + row2) * (1/64)   rows sy = y-1, y, y+1
 *   dst(2y+1) = ((row1 + row2)*4) * (1/64)
 */
static void pyr_up_hrow(const double *s, int cols, double *row)
{
    if (cols == 1) {
        row[0] = row[1] = s[0] * 8;
        return;
    }
    row[0] = s[0] * 6 + s[1] * 2;
    row[1] = (s[0] + s[1]) * 4;
    row[2 * (cols - 1)] = s[cols - 2] + s[cols - 1] * 7;
    row[2 * (cols - 1) + 1] = s[cols - 1] * 8;
    for (int x = 1; x < cols - 1; ++x) {
        row[2 * x] = s[x - 1] + s[x] * 6 + s[x + 1];
        row[2 * x + 1] = (s[x] + s[x + 1]) * 4;
    }
}

void lko_pyr_up(const double *src, int rows, int cols, double *dst)
{
    int dcols = 2 * cols, drows = 2 * rows;
    double *hbuf = (double *)malloc((size_t)rows * dcols * sizeof(double));
    for (int y = 0; y < rows; ++y) pyr_up_hrow(src + (size_t)y * cols, cols, hbuf + (size_t)y * dcols);
    for (int y = 0; y < rows; ++y) {
        const double *row0 = hbuf + (size_t)(r101(2 * (y - 1), drows) / 2) * dcols;
        const double *row1 = hbuf + (size_t)(r101(2 * y, drows) / 2) * dcols;
        const double *row2 = hbuf + (size_t)(r101(2 * (y + 1), drows) / 2) * dcols;
        double *d0 = dst + (size_t)(2 * y) * dcols;
        double *d1 = dst + (size_t)(2 * y + 1) * dcols;
        for (int x = 0; x < dcols; ++x) {
            double t1 = ((row1[x] + row2[x]) * 4) * (1.0 / 64);
            double t0 = (row0[x] + row1[x] * 6 + row2[x]) * (1.0 / 64);
            d1[x] = t1;
            d0[x] = t0;
        }
    }
    free(hbuf);
}

/* SerialPyr.cpp:181-187: f = Point2f((float)upu,(float)upv); map = Point2f(x +/- f.x/2, y +/- f.y/2)
 * -- float32 arithmetic throughout (int x converts to float). */
void lko_build_map(const double *upu, const double *upv, int rows, int cols, float sign,
                   float *map_xy)
{
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            size_t i = (size_t)y * cols + x;
            float fx = (float)upu[i], fy = (float)upv[i];
            float hx = fx / 2, hy = fy / 2;
            if (sign > 0) {
                map_xy[2 * i] = (float)x + hx;
                map_xy[2 * i + 1] = (float)y + hy;
            } else {
                map_xy[2 * i] = (float)x - hx;
                map_xy[2 * i + 1] = (float)y - hy;
            }
        }
}

/* cvRound as OpenCV 2.4 compiles it on x86-64 (cvtsd2si: round-half-even, out-of-range/NaN ->
 * 0x80000000 "integer indefinite"). */
static int cv_round(double v)
{
    if (!(v > -2147483648.5 && v < 2147483647.5)) return INT_MIN;
    return (int)nearbyint(v);
}

static short sat_short(int v)
{
    return (short)(v < SHRT_MIN ? SHRT_MIN : (v > SHRT_MAX ? SHRT_MAX : v));
}

/*
 * cv::remap(src, dst, map(CV_32FC2), Mat(), INTER_LINEAR) with BORDER_CONSTANT, value 0.
 * OpenCV 2.4 imgwarp.cpp: RemapInvoker converts the float map to fixed point
 *   sx = cvRound(mx*32), sy = cvRound(my*32); ix = sat<short>(sx>>5), iy = sat<short>(sy>>5);
 *   a = sx&31, b = sy&31
 * and remapBilinear<Cast<double,double>,RemapNoVec,float> blends with the float table
 *   w = {(1-b')(1-a'), (1-b')a', b'(1-a'), b'a'},  a' = a/32, b' = b/32 (exact in float)
 *   D = S00*w0 + S01*w1 + S10*w2 + S11*w3          (fp64 accumulation, left to right)
 * Taps outside the source contribute the border value 0.
 */
void lko_remap_linear(const double *src, int rows, int cols, const float *map_xy, double *dst)
{
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            size_t i = (size_t)y * cols + x;
            float mx = map_xy[2 * i] * 32.0f, my = map_xy[2 * i + 1] * 32.0f;
            int sx = cv_round((double)mx), sy = cv_round((double)my);
            int ix = sat_short(sx >> 5), iy = sat_short(sy >> 5);
            float a = (float)(sx & 31) * (1.0f / 32), b = (float)(sy & 31) * (1.0f / 32);
            float w0 = (1.0f - b) * (1.0f - a), w1 = (1.0f - b) * a, w2 = b * (1.0f - a), w3 = b * a;
            int x0ok = ix >= 0 && ix < cols, x1ok = ix + 1 >= 0 && ix + 1 < cols;
            int y0ok = iy >= 0 && iy < rows, y1ok = iy + 1 >= 0 && iy + 1 < rows;
            double v0 = (y0ok && x0ok) ? src[(size_t)iy * cols + ix] : 0.0;
            double v1 = (y0ok && x1ok) ? src[(size_t)iy * cols + ix + 1] : 0.0;
            double v2 = (y1ok && x0ok) ? src[(size_t)(iy + 1) * cols + ix] : 0.0;
            double v3 = (y1ok && x1ok) ? src[(size_t)(iy + 1) * cols + ix + 1] : 0.0;
            dst[i] = v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3;
        }
}

/* SerialPyr.cpp:157-194 */
int lko_coarse_to_fine(double *img1, double *img2, int rows, int cols, int levels, int win,
                       double *u, double *v)
{
    if (levels < 1 || levels > 32 || levels > lko_max_layer(cols, rows)) return -1;
    if (win < 3 || (win & 1) == 0) return -2;

    double *pyr1[32], *pyr2[32];
    int prow[32], pcol[32];
    pyr1[0] = img1;                 /* :148 push_back(img): header share, no copy */
    pyr2[0] = img2;
    prow[0] = rows; pcol[0] = cols;
    for (int l = 1; l < levels; ++l) {            /* :146-155 */
        prow[l] = (prow[l - 1] + 1) / 2;
        pcol[l] = (pcol[l - 1] + 1) / 2;
        size_t n = (size_t)prow[l] * pcol[l];
        pyr1[l] = (double *)malloc(n * sizeof(double));
        pyr2[l] = (double *)malloc(n * sizeof(double));
        lko_pyr_down(pyr1[l - 1], prow[l - 1], pcol[l - 1], pyr1[l]);
        lko_pyr_down(pyr2[l - 1], prow[l - 1], pcol[l - 1], pyr2[l]);
    }

    double *upu = NULL, *upv = NULL;
    for (int i = levels - 1; i >= 0; --i) {       /* :162 */
        size_t n = (size_t)prow[i] * pcol[i];
        double *tmpu = (i == 0) ? u : (double *)malloc(n * sizeof(double));
        double *tmpv = (i == 0) ? v : (double *)malloc(n * sizeof(double));
        lko_lucas_kanade(pyr1[i], pyr2[i], prow[i], pcol[i], win, tmpu, tmpv);   /* :166 */
        if (i != levels - 1) {                    /* :167-170 */
            for (size_t k = 0; k < n; ++k) { tmpu[k] += upu[k]; tmpv[k] += upv[k]; }
        }
        free(upu); free(upv); upu = upv = NULL;
        if (i == 0) break;                        /* :171-175 */

        int r2 = prow[i - 1], c2 = pcol[i - 1];   /* == 2*prow[i], 2*pcol[i] (levels <= max_layer) */
        size_t n2 = (size_t)r2 * c2;
        upu = (double *)malloc(n2 * sizeof(double));
        upv = (double *)malloc(n2 * sizeof(double));
        lko_pyr_up(tmpu, prow[i], pcol[i], upu);  /* :176-177 */
        lko_pyr_up(tmpv, prow[i], pcol[i], upv);
        free(tmpu); free(tmpv);

        float *map = (float *)malloc(2 * n2 * sizeof(float));
        double *warped = (double *)malloc(n2 * sizeof(double));
        lko_build_map(upu, upv, r2, c2, +1.0f, map);                    /* :184 map1 */
        lko_remap_linear(pyr1[i - 1], r2, c2, map, warped);             /* :189 */
        memcpy(pyr1[i - 1], warped, n2 * sizeof(double));               /* :191 copyTo (in place) */
        lko_build_map(upu, upv, r2, c2, -1.0f, map);                    /* :185 map2 */
        lko_remap_linear(pyr2[i - 1], r2, c2, map, warped);             /* :190 */
        memcpy(pyr2[i - 1], warped, n2 * sizeof(double));               /* :192 */
        free(map); free(warped);
    }
    for (int l = 1; l < levels; ++l) { free(pyr1[l]); free(pyr2[l]); }
    return 0;
}

/* SerialPyr.cpp:237 convertTo(CV_64FC1, 1.0/255, 0): saturate_cast<double>(src*alpha + beta) */
void lko_u8_to_f64(const unsigned char *src, long n, double *dst)
{
    for (long i = 0; i < n; ++i) dst[i] = (double)src[i] * (1.0 / 255);
}

/* ParallelVideoPyr.cpp:850-873: val = sqrt(u*u+v*v); pixels with val >= thresh are counted and
 * their column (x) / row (y) indices summed (avgX += j; avgY += i; counts++). */
void lko_flow_centroid(const double *u, const double *v, int rows, int cols, double thresh,
                       long long *count, double *sum_x, double *sum_y)
{
    long long c = 0;
    double sx = 0, sy = 0;
    for (int i = 0; i < rows; ++i)
        for (int j = 0; j < cols; ++j) {
            double xf = u[(size_t)i * cols + j], yf = v[(size_t)i * cols + j];
            double val = sqrt(xf * xf + yf * yf);
            if (!(val < thresh)) { sx += j; sy += i; ++c; }
        }
    *count = c; *sum_x = sx; *sum_y = sy;
}

/*
 * ParallelVideoPyr.cpp:803-813.  OpenCV 2.4 semantics restated:
 *   threshold(.., THRESH_TOZERO): v > thresh ? v : 0                                   (thresh.cpp)
 *   Sobel(src, dst, CV_64F, 1, 0): sepFilter2D with kx = [-1 0 1], ky = [1 2 1]; (0, 1): kx = [1 2 1],
 *     ky = [-1 0 1]; correlation, BORDER_DEFAULT = REFLECT_101                       (deriv.cpp)
 *   dilate / erode(src, dst, Mat(), Point(-1,-1), 2): 3x3 rectangle, two iterations, BORDER_CONSTANT with
 *     morphologyDefaultBorderValue() = "pixels outside the image never win"           (morph.cpp)
 */
static void morph3x3(const double *src, int rows, int cols, int is_dilate, double *dst)
{
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            double m = src[(size_t)y * cols + x];
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    int yy = y + dy, xx = x + dx;
                    if (yy < 0 || yy >= rows || xx < 0 || xx >= cols) continue;
                    double v = src[(size_t)yy * cols + xx];
                    if (is_dilate ? v > m : v < m) m = v;
                }
            dst[(size_t)y * cols + x] = m;
        }
}

void lko_edge_map_u8(const unsigned char *diff, int rows, int cols, int thresh, double *out)
{
    size_t n = (size_t)rows * cols;
    double *t = (double *)malloc(n * sizeof(double)), *e = (double *)malloc(n * sizeof(double));
    for (size_t i = 0; i < n; ++i) t[i] = diff[i] > thresh ? (double)diff[i] : 0.0;   /* :803 */
    for (int y = 0; y < rows; ++y) {                                                   /* :808-811 */
        int ym = r101(y - 1, rows), yp = r101(y + 1, rows);
        for (int x = 0; x < cols; ++x) {
            int xm = r101(x - 1, cols), xp = r101(x + 1, cols);
#define T(yy, xx) t[(size_t)(yy) * cols + (xx)]
            double sx = (T(ym, xp) - T(ym, xm)) + 2 * (T(y, xp) - T(y, xm)) + (T(yp, xp) - T(yp, xm));
            double sy = (T(yp, xm) - T(ym, xm)) + 2 * (T(yp, x) - T(ym, x)) + (T(yp, xp) - T(ym, xp));
#undef T
            e[(size_t)y * cols + x] = sx + sy;             /* exact integers: summation order is irrelevant */
        }
    }
    morph3x3(e, rows, cols, 1, t);                          /* :812 dilate, iterations = 2 */
    morph3x3(t, rows, cols, 1, e);
    morph3x3(e, rows, cols, 0, t);                          /* :813 erode, iterations = 2 */
    morph3x3(t, rows, cols, 0, out);
    free(t); free(e);
}

/*
 * ParallelVideoPyr.cpp:786-787 -- GaussianBlur(frame, frame, Size(k,k), sigma, sigma) on the 8UC3 frame, then
 * cvtColor(.., CV_BGR2GRAY).  PARITY UNPINNED: the reference holds no fixture for its pre-processing chain
 * and OpenCV is absent, so this restates OpenCV 2.4's published 8-bit algorithms:
 *   getGaussianKernel(n, sigma, CV_32F) (smooth.cpp): cf[i] = (float)exp(-0.5/sigma^2 * x^2), sum over the
 *     floats, cf[i] = (float)(cf[i] * (1/sum));
 *   createSeparableLinearFilter (filter.cpp), 8U -> 8U with a smooth symmetric kernel: fixed point, bits = 8:
 *     integer kernel = cvRound(cf[i] * 256); row pass 8U -> 32S; column pass (sum of 16-bit-scaled products)
 *     rounded to 8U.  The x86 SSE2 column filter (SymmColumnVec_32s8u, what the reference's x86-64 binary
 *     runs for every column of a 1080-wide BGR row) evaluates the exact sum in float and converts with
 *     round-half-to-EVEN; the scalar tail (FixedPtCastEx) would round half up.  The two differ only on exact
 *     ties; this restatement follows the SSE2 build.  BORDER_DEFAULT = reflect-101.
 *   cvtColor BGR2GRAY 8U (color.cpp RGB2Gray<uchar>): (B*1868 + G*9617 + R*4899 + 8192) >> 14.
 * For (9, 1.5) the integer kernel is [2 9 28 55 68 55 28 9 2] (sum 256).
 */
void lko_gaussian_kernel_q8(int n, double sigma, int *k)
{
    float cf[64];
    double scale2x = -0.5 / (sigma * sigma), sum = 0.0;
    for (int i = 0; i < n; ++i) {
        double x = i - (n - 1) * 0.5;
        cf[i] = (float)exp(scale2x * x * x);
        sum += cf[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < n; ++i) {
        cf[i] = (float)(cf[i] * sum);
        k[i] = (int)lrint((double)(cf[i] * 256.0f));     /* convertTo(CV_32S, 1 << 8): cvRound of the float product */
    }
}

void lko_blur_gray_bgr8(const unsigned char *bgr, int rows, int cols, int ksize, double sigma,
                        unsigned char *gray)
{
    int k[64];
    const int r = ksize / 2;
    lko_gaussian_kernel_q8(ksize, sigma, k);
    int *row = (int *)malloc((size_t)rows * cols * 3 * sizeof(int));
    for (int y = 0; y < rows; ++y)                           /* row pass: 8U -> 32S */
        for (int x = 0; x < cols; ++x)
            for (int c = 0; c < 3; ++c) {
                int s = 0;
                for (int j = -r; j <= r; ++j)
                    s += k[j + r] * bgr[((size_t)y * cols + r101(x + j, cols)) * 3 + c];
                row[((size_t)y * cols + x) * 3 + c] = s;
            }
    for (int y = 0; y < rows; ++y)                           /* column pass + BGR2GRAY */
        for (int x = 0; x < cols; ++x) {
            int px[3];
            for (int c = 0; c < 3; ++c) {
                long long s = 0;
                for (int j = -r; j <= r; ++j)
                    s += (long long)k[j + r] * row[((size_t)r101(y + j, rows) * cols + x) * 3 + c];
                long long q = s >> 16, rem = s & 0xFFFF;     /* s / 65536 rounded half to even */
                if (rem > 0x8000 || (rem == 0x8000 && (q & 1))) ++q;
                px[c] = q < 0 ? 0 : (q > 255 ? 255 : (int)q);
            }
            gray[(size_t)y * cols + x] = (unsigned char)((px[0] * 1868 + px[1] * 9617 + px[2] * 4899 + 8192) >> 14);
        }
    free(row);
}

/*
 * ParallelVideoPyr.cpp:800 -- Mat diff = current_frame - LEARNING_RATE * prevFrame on 8UC1 frames.
 * PARITY UNPINNED (no fixture).  OpenCV 2.4: the MatExpr lowers to addWeighted(cur, 1, prev, -rate, 0)
 * (matop.cpp MatOp_AddEx::assign), whose 8-bit kernel (arithm.cpp addWeighted8u, SSE2 and scalar alike)
 * computes in FLOAT: t = (float)cur*alpha + (float)prev*beta + gamma with alpha, beta, gamma cast to float,
 * every product and sum rounded to float, then saturate_cast<uchar>(cvRound(t)) (round half to even).
 */
void lko_frame_diff_u8(const unsigned char *cur, const unsigned char *prev, long n, double rate,
                       unsigned char *diff)
{
    const float alpha = 1.0f, beta = (float)(-rate), gamma = 0.0f;
    for (long i = 0; i < n; ++i) {
        volatile float a = (float)cur[i] * alpha;            /* volatile: every step rounded to float */
        volatile float b = (float)prev[i] * beta;
        volatile float t = a + b;
        t = t + gamma;
        long v = lrintf(t);
        diff[i] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

/*
 * ParallelVideoPyr.cpp:778 -- resize(frame, current_frame, Size(1080, 1080), 0, 0, INTER_CUBIC) on the 8UC3 frame.
 * PARITY UNPINNED: the reference holds no fixture for its pre-processing chain and OpenCV is absent, so this
 * restates OpenCV 2.4's published algorithm (imgwarp.cpp: cv::resize -> resizeGeneric_<HResizeCubic<uchar,int,short>,
 * VResizeCubic<uchar,int,short,FixedPtCast<int,uchar,22>,VResizeCubicVec_32s8u>>):
 *   scale = 1 / ((double)dsize / ssize); per destination index d: f = (float)((d + 0.5) * scale - 0.5), s = floor(f),
 *     f -= s; the four cubic weights (A = -0.75, float arithmetic, interpolateCubic) are stored as
 *     saturate_cast<short>(w * 2048) (cvRound: half to even);
 *   horizontal pass 8U -> 32S over the taps s-1 .. s+2 of each channel, indices outside the row clamped to its ends;
 *   vertical pass over the rows s-1 .. s+2 (clamped): the x86 SSE2 vector op (what the reference's x86-64 binary runs)
 *     takes the row elements 8 at a time, in FLOAT -- ((S0*b0 + S1*b1) + S2*b2) + S3*b3 with b = beta / 2^22, every
 *     product and sum rounded to float, cvtps2dq (half to even), saturated to 0..255; the last (width*3) mod 8 elements of
 *     a row take the scalar path (S0*b0 + S1*b1 + S2*b2 + S3*b3 + 2^21) >> 22 in 32-bit integers.
 */
static void cubic_tab(int ssize, int dsize, int *ofs, short *coef)
{
    const double scale = 1.0 / ((double)dsize / ssize);
    for (int d = 0; d < dsize; ++d) {
        volatile float f = (float)((d + 0.5) * scale - 0.5);
        const int s = (int)floorf(f);
        f = f - (float)s;
        const float A = -0.75f, x = f;
        volatile float c[4], t;                              /* volatile: every step rounded to float, nothing fused */
        t = A * (x + 1.0f); t = t - 5.0f * A; t = t * (x + 1.0f); t = t + 8.0f * A; t = t * (x + 1.0f); c[0] = t - 4.0f * A;
        t = (A + 2.0f) * x; t = t - (A + 3.0f); t = t * x; t = t * x; c[1] = t + 1.0f;
        t = (A + 2.0f) * (1.0f - x); t = t - (A + 3.0f); t = t * (1.0f - x); t = t * (1.0f - x); c[2] = t + 1.0f;
        t = 1.0f - c[0]; t = t - c[1]; c[3] = t - c[2];
        ofs[d] = s;
        for (int k = 0; k < 4; ++k) {
            long v = lrintf(c[k] * 2048.0f);                 /* INTER_RESIZE_COEF_SCALE = 1 << 11 */
            coef[4 * d + k] = (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
        }
    }
}

void lko_resize_cubic_tables(int ssize, int dsize, int *ofs, short *coef) { cubic_tab(ssize, dsize, ofs, coef); }

void lko_resize_cubic_bgr8(const unsigned char *src, int srows, int scols, int drows, int dcols, unsigned char *dst)
{
    int *xofs = (int *)malloc((size_t)dcols * sizeof(int)), *yofs = (int *)malloc((size_t)drows * sizeof(int));
    short *alpha = (short *)malloc((size_t)dcols * 4 * sizeof(short)), *beta = (short *)malloc((size_t)drows * 4 * sizeof(short));
    int *rows4 = (int *)malloc((size_t)4 * dcols * 3 * sizeof(int));
    cubic_tab(scols, dcols, xofs, alpha);
    cubic_tab(srows, drows, yofs, beta);
    const int width = dcols * 3, vec_end = width & ~7;
    const float scale = 1.0f / (2048.0f * 2048.0f);
    for (int dy = 0; dy < drows; ++dy) {
        for (int k = 0; k < 4; ++k) {                        /* horizontal pass of the four source rows */
            int sy = yofs[dy] - 1 + k;
            sy = sy < 0 ? 0 : (sy > srows - 1 ? srows - 1 : sy);
            const unsigned char *S = src + (size_t)sy * scols * 3;
            int *D = rows4 + (size_t)k * width;
            for (int dx = 0; dx < dcols; ++dx)
                for (int c = 0; c < 3; ++c) {
                    int v = 0;
                    for (int j = 0; j < 4; ++j) {
                        int sx = xofs[dx] - 1 + j;
                        sx = sx < 0 ? 0 : (sx > scols - 1 ? scols - 1 : sx);
                        v += S[sx * 3 + c] * alpha[4 * dx + j];
                    }
                    D[dx * 3 + c] = v;
                }
        }
        const short *b = beta + 4 * dy;
        const int *S0 = rows4, *S1 = rows4 + width, *S2 = rows4 + 2 * width, *S3 = rows4 + 3 * width;
        const float b0 = b[0] * scale, b1 = b[1] * scale, b2 = b[2] * scale, b3 = b[3] * scale;
        unsigned char *o = dst + (size_t)dy * width;
        for (int x = 0; x < width; ++x) {
            long v;
            if (x < vec_end) {
                volatile float s = (float)S0[x] * b0, f = (float)S1[x] * b1;
                s = s + f; f = (float)S2[x] * b2; s = s + f; f = (float)S3[x] * b3; s = s + f;
                v = lrintf(s);
            } else {
                v = (S0[x] * b[0] + S1[x] * b[1] + S2[x] * b[2] + S3[x] * b[3] + (1 << 21)) >> 22;
            }
            o[x] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
    free(rows4); free(beta); free(alpha); free(yofs); free(xofs);
}
