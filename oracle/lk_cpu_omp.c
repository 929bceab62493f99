/*
 * lk_cpu_omp.c -- "restated reference CPU path": the CPU baseline that bench.py times next to the
 * GPU numbers (cpu_baseline.kind = "port").  TEST/BENCH INFRASTRUCTURE ONLY, like lk_oracle.c.
 *
 * The reference binary cannot be built here (needs OpenCV 2.4, absent), so this
 * file restates /root/reference/ParallelVideoPyr.cpp's hot call with the SAME parallel structure:
 *   :587-597  2-way task : Gaussian pyramid of image 1 || image 2
 *   :347-364  3-way task : get_fx || get_fy || get_ft
 *   :383-405  5-way task : the five element-wise products
 *   :162-191  omp_set_num_threads(8) + "parallel for" over rows : get_Sum9_Mat, called 5x serially
 *   :470-472  serial      : det and the two numerators
 *   :510-520  2-way task : the two divides
 *   :667-684  parallel for over rows : map build
 *   :691-703  2-way task : remap + copyTo of image 1 || image 2
 * with full-plane temporaries between stages (mode 0).  The reference's
 * "if(omp_get_thread_num()==k)" idiom silently skips work when fewer threads are granted; this
 * restatement uses "omp sections", which computes the same thing with any thread count.
 * mode 1 = same arithmetic, every stage row-parallel over all cores (so that the GPU/CPU ratio is
 * not flattered by the 2/3/5-way task limits).
 *
 * Arithmetic is identical to lk_oracle.c (tests check bit-equality of the two).
 */
#include "lk_oracle.h"

#include <limits.h>
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

static int r101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

static int max_layer(int cols, int rows)
{
    int p = 1, res;
    if (cols <= 0 || rows <= 0) return 0;
    while (cols % (1 << p) == 0) ++p;
    res = p;
    p = 1;
    while (rows % (1 << p) == 0) ++p;
    return res < p ? res : p;
}

/* ---- stage kernels over a row range [y0,y1) ------------------------------------------------ */

static void filter2d_rows(const double *src, int rows, int cols, double k00, double k01, double k10,
                          double k11, double *dst, int y0, int y1)
{
    for (int y = y0; y < y1; ++y) {
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

/* one derivative plane = two filter2D + add (ParallelVideoPyr.cpp:44-84), rows [y0,y1) */
static void deriv_rows(const double *i1, const double *i2, int rows, int cols, const double k1[4],
                       const double k2[4], double *d1, double *d2, double *out, int y0, int y1)
{
    filter2d_rows(i1, rows, cols, k1[0], k1[1], k1[2], k1[3], d1, y0, y1);
    filter2d_rows(i2, rows, cols, k2[0], k2[1], k2[2], k2[3], d2, y0, y1);
    for (size_t i = (size_t)y0 * cols; i < (size_t)y1 * cols; ++i) out[i] = d1[i] + d2[i];
}

static void mul_rows(const double *a, const double *b, double *out, size_t i0, size_t i1)
{
    for (size_t i = i0; i < i1; ++i) out[i] = a[i] * b[i];
}

/* get_Sum9 (ParallelVideoPyr.cpp:114-160) widened to win x win; bounds-checked per tap like the
 * reference */
static double sum_win_at(const double *m, int rows, int cols, int y, int x, int r)
{
    double val = 0.0;
    int tmp = 0;
    for (int i = -r; i <= r; ++i)
        for (int j = -r; j <= r; ++j) {
            int yy = y + i, xx = x + j;
            if (xx >= 0 && xx < cols && yy >= 0 && yy < rows) {
                ++tmp;
                val += m[(size_t)yy * cols + xx];
            }
        }
    int full = (2 * r + 1) * (2 * r + 1);
    return tmp == full ? val : m[(size_t)y * cols + x] * full;
}

/* get_Sum9_Mat (ParallelVideoPyr.cpp:162-191): zeros + row-parallel loop with nthr threads */
static void sum_win_mat(const double *m, int rows, int cols, int win, double *dst, int nthr)
{
    int r = win / 2;
    memset(dst, 0, (size_t)rows * cols * sizeof(double));
#pragma omp parallel for num_threads(nthr) schedule(static)
    for (int i = 1; i < rows - 1; ++i)
        for (int j = 1; j < cols - 1; ++j)
            dst[(size_t)i * cols + j] = sum_win_at(m, rows, cols, i, j, r);
}

static void pyr_down(const double *src, int rows, int cols, double *dst, int par)
{
    int drows = (rows + 1) / 2, dcols = (cols + 1) / 2;
    double *hbuf = (double *)malloc((size_t)rows * dcols * sizeof(double));
#pragma omp parallel for if (par) schedule(static)
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
#pragma omp parallel for if (par) schedule(static)
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

static void pyr_up_hrow(const double *s, int cols, double *row)
{
    if (cols == 1) { row[0] = row[1] = s[0] * 8; return; }
    row[0] = s[0] * 6 + s[1] * 2;
    row[1] = (s[0] + s[1]) * 4;
    row[2 * (cols - 1)] = s[cols - 2] + s[cols - 1] * 7;
    row[2 * (cols - 1) + 1] = s[cols - 1] * 8;
    for (int x = 1; x < cols - 1; ++x) {
        row[2 * x] = s[x - 1] + s[x] * 6 + s[x + 1];
        row[2 * x + 1] = (s[x] + s[x + 1]) * 4;
    }
}

/* cv::pyrUp is serial in the reference (OpenCV 2.4 has no parallel pyrUp); par only in mode 1 */
static void pyr_up(const double *src, int rows, int cols, double *dst, int par)
{
    int dcols = 2 * cols, drows = 2 * rows;
    double *hbuf = (double *)malloc((size_t)rows * dcols * sizeof(double));
#pragma omp parallel for if (par) schedule(static)
    for (int y = 0; y < rows; ++y) pyr_up_hrow(src + (size_t)y * cols, cols, hbuf + (size_t)y * dcols);
#pragma omp parallel for if (par) schedule(static)
    for (int y = 0; y < rows; ++y) {
        const double *row0 = hbuf + (size_t)(r101(2 * (y - 1), drows) / 2) * dcols;
        const double *row1 = hbuf + (size_t)(r101(2 * y, drows) / 2) * dcols;
        const double *row2 = hbuf + (size_t)(r101(2 * (y + 1), drows) / 2) * dcols;
        double *d0 = dst + (size_t)(2 * y) * dcols;
        double *d1 = dst + (size_t)(2 * y + 1) * dcols;
        for (int x = 0; x < dcols; ++x) {
            d1[x] = ((row1[x] + row2[x]) * 4) * (1.0 / 64);
            d0[x] = (row0[x] + row1[x] * 6 + row2[x]) * (1.0 / 64);
        }
    }
    free(hbuf);
}

static int cv_round(double v)
{
    if (!(v > -2147483648.5 && v < 2147483647.5)) return INT_MIN;
    return (int)nearbyint(v);
}
static short sat_short(int v) { return (short)(v < SHRT_MIN ? SHRT_MIN : (v > SHRT_MAX ? SHRT_MAX : v)); }

static void remap_rows(const double *src, int rows, int cols, const float *map_xy, double *dst,
                       int y0, int y1)
{
    for (int y = y0; y < y1; ++y)
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

/* ---- getLucasKanadeOpticalFlow (ParallelVideoPyr.cpp:327-542) ------------------------------- */

static const double KX[4] = {-1, 1, -1, 1}, KY[4] = {-1, -1, 1, 1}, KM[4] = {-1, -1, -1, -1},
                    KP[4] = {1, 1, 1, 1};

static void lucas_kanade(const double *i1, const double *i2, int rows, int cols, int win, int mode,
                         double *u, double *v)
{
    size_t n = (size_t)rows * cols;
    double *buf = (double *)malloc(20 * n * sizeof(double));
    double *fx = buf, *fy = buf + n, *ft = buf + 2 * n;
    double *p[5], *s[5], *scr[6];
    for (int k = 0; k < 5; ++k) { p[k] = buf + (3 + k) * n; s[k] = buf + (8 + k) * n; }
    for (int k = 0; k < 6; ++k) scr[k] = buf + (13 + k) * n;
    double *tmp = buf + 19 * n;

    if (mode == 0) {
#pragma omp parallel sections num_threads(3)      /* :347-364 */
        {
#pragma omp section
            deriv_rows(i1, i2, rows, cols, KX, KX, scr[0], scr[1], fx, 0, rows);
#pragma omp section
            deriv_rows(i1, i2, rows, cols, KY, KY, scr[2], scr[3], fy, 0, rows);
#pragma omp section
            deriv_rows(i1, i2, rows, cols, KM, KP, scr[4], scr[5], ft, 0, rows);
        }
#pragma omp parallel sections num_threads(5)      /* :383-405 */
        {
#pragma omp section
            mul_rows(fx, fx, p[0], 0, n);          /* fx2  */
#pragma omp section
            mul_rows(fy, fy, p[1], 0, n);          /* fy2  */
#pragma omp section
            mul_rows(fx, fy, p[2], 0, n);          /* fxfy */
#pragma omp section
            mul_rows(fx, ft, p[3], 0, n);          /* fxft */
#pragma omp section
            mul_rows(fy, ft, p[4], 0, n);          /* fyft */
        }
    } else {
#pragma omp parallel for schedule(static)
        for (int y = 0; y < rows; ++y) {
            deriv_rows(i1, i2, rows, cols, KX, KX, scr[0], scr[1], fx, y, y + 1);
            deriv_rows(i1, i2, rows, cols, KY, KY, scr[2], scr[3], fy, y, y + 1);
            deriv_rows(i1, i2, rows, cols, KM, KP, scr[4], scr[5], ft, y, y + 1);
            size_t a = (size_t)y * cols, b = a + cols;
            mul_rows(fx, fx, p[0], a, b);
            mul_rows(fy, fy, p[1], a, b);
            mul_rows(fx, fy, p[2], a, b);
            mul_rows(fx, ft, p[3], a, b);
            mul_rows(fy, ft, p[4], a, b);
        }
    }

    int nthr = mode == 0 ? 8 : omp_get_max_threads();      /* :164-165 omp_set_num_threads(8) */
    for (int k = 0; k < 5; ++k) sum_win_mat(p[k], rows, cols, win, s[k], nthr);   /* :442-446 */
    const double *sxx = s[0], *syy = s[1], *sxy = s[2], *sxt = s[3], *syt = s[4];

    if (mode == 0) {
        for (size_t i = 0; i < n; ++i) {          /* :470-472 serial */
            tmp[i] = sxx[i] * syy[i] - sxy[i] * sxy[i];
            u[i] = sxy[i] * syt[i] - syy[i] * sxt[i];
            v[i] = sxt[i] * sxy[i] - sxx[i] * syt[i];
        }
#pragma omp parallel sections num_threads(2)      /* :510-520 */
        {
#pragma omp section
            for (size_t i = 0; i < n; ++i) u[i] = tmp[i] != 0 ? u[i] / tmp[i] : 0.0;
#pragma omp section
            for (size_t i = 0; i < n; ++i) v[i] = tmp[i] != 0 ? v[i] / tmp[i] : 0.0;
        }
    } else {
#pragma omp parallel for schedule(static)
        for (int y = 0; y < rows; ++y)
            for (size_t i = (size_t)y * cols; i < (size_t)(y + 1) * cols; ++i) {
                double d = sxx[i] * syy[i] - sxy[i] * sxy[i];
                double un = sxy[i] * syt[i] - syy[i] * sxt[i];
                double vn = sxt[i] * sxy[i] - sxx[i] * syt[i];
                u[i] = d != 0 ? un / d : 0.0;
                v[i] = d != 0 ? vn / d : 0.0;
            }
    }
    free(buf);
}

/* ---- coarseToFineEstimation (ParallelVideoPyr.cpp:578-725) ---------------------------------- */

int lkc_num_threads(void) { return omp_get_max_threads(); }
/* mode 1 (all-cores row loops) uses omp_get_max_threads(): let the caller pin it to the cores it may really use */
void lkc_set_num_threads(int n) { if (n > 0) omp_set_num_threads(n); }

int lkc_coarse_to_fine(double *img1, double *img2, int rows, int cols, int levels, int win, int mode,
                       double *u, double *v)
{
    if (levels < 1 || levels > 32 || levels > max_layer(cols, rows)) return -1;
    if (win < 3 || (win & 1) == 0) return -2;
    double *pyr1[32], *pyr2[32];
    int prow[32], pcol[32];
    pyr1[0] = img1; pyr2[0] = img2; prow[0] = rows; pcol[0] = cols;
    for (int l = 1; l < levels; ++l) {
        prow[l] = (prow[l - 1] + 1) / 2;
        pcol[l] = (pcol[l - 1] + 1) / 2;
        pyr1[l] = (double *)malloc((size_t)prow[l] * pcol[l] * sizeof(double));
        pyr2[l] = (double *)malloc((size_t)prow[l] * pcol[l] * sizeof(double));
    }
    if (mode == 0) {
#pragma omp parallel sections num_threads(2)      /* :587-597 */
        {
#pragma omp section
            for (int l = 1; l < levels; ++l) pyr_down(pyr1[l - 1], prow[l - 1], pcol[l - 1], pyr1[l], 0);
#pragma omp section
            for (int l = 1; l < levels; ++l) pyr_down(pyr2[l - 1], prow[l - 1], pcol[l - 1], pyr2[l], 0);
        }
    } else {
        for (int l = 1; l < levels; ++l) {
            pyr_down(pyr1[l - 1], prow[l - 1], pcol[l - 1], pyr1[l], 1);
            pyr_down(pyr2[l - 1], prow[l - 1], pcol[l - 1], pyr2[l], 1);
        }
    }

    double *upu = NULL, *upv = NULL;
    for (int i = levels - 1; i >= 0; --i) {
        size_t n = (size_t)prow[i] * pcol[i];
        double *tmpu = (i == 0) ? u : (double *)malloc(n * sizeof(double));
        double *tmpv = (i == 0) ? v : (double *)malloc(n * sizeof(double));
        lucas_kanade(pyr1[i], pyr2[i], prow[i], pcol[i], win, mode, tmpu, tmpv);   /* :621 */
        if (i != levels - 1)                                                      /* :634-637 */
            for (size_t k = 0; k < n; ++k) { tmpu[k] += upu[k]; tmpv[k] += upv[k]; }
        free(upu); free(upv); upu = upv = NULL;
        if (i == 0) break;

        int r2 = prow[i - 1], c2 = pcol[i - 1];
        size_t n2 = (size_t)r2 * c2;
        upu = (double *)malloc(n2 * sizeof(double));
        upv = (double *)malloc(n2 * sizeof(double));
        pyr_up(tmpu, prow[i], pcol[i], upu, mode == 1);                           /* :646-647 */
        pyr_up(tmpv, prow[i], pcol[i], upv, mode == 1);
        free(tmpu); free(tmpv);

        float *map1 = (float *)malloc(2 * n2 * sizeof(float));
        float *map2 = (float *)malloc(2 * n2 * sizeof(float));
#pragma omp parallel for schedule(static)          /* :667-684 */
        for (int y = 0; y < r2; ++y)
            for (int x = 0; x < c2; ++x) {
                size_t k = (size_t)y * c2 + x;
                float fx = (float)upu[k], fy = (float)upv[k];
                map1[2 * k] = (float)x + fx / 2;
                map1[2 * k + 1] = (float)y + fy / 2;
                map2[2 * k] = (float)x - fx / 2;
                map2[2 * k + 1] = (float)y - fy / 2;
            }
        double *w1 = (double *)malloc(n2 * sizeof(double));
        double *w2 = (double *)malloc(n2 * sizeof(double));
        if (mode == 0) {
#pragma omp parallel sections num_threads(2)      /* :691-703 */
            {
#pragma omp section
                { remap_rows(pyr1[i - 1], r2, c2, map1, w1, 0, r2); memcpy(pyr1[i - 1], w1, n2 * sizeof(double)); }
#pragma omp section
                { remap_rows(pyr2[i - 1], r2, c2, map2, w2, 0, r2); memcpy(pyr2[i - 1], w2, n2 * sizeof(double)); }
            }
        } else {
#pragma omp parallel for schedule(static)
            for (int y = 0; y < r2; ++y) {
                remap_rows(pyr1[i - 1], r2, c2, map1, w1, y, y + 1);
                remap_rows(pyr2[i - 1], r2, c2, map2, w2, y, y + 1);
            }
            memcpy(pyr1[i - 1], w1, n2 * sizeof(double));
            memcpy(pyr2[i - 1], w2, n2 * sizeof(double));
        }
        free(map1); free(map2); free(w1); free(w2);
    }
    for (int l = 1; l < levels; ++l) { free(pyr1[l]); free(pyr2[l]); }
    return 0;
}
