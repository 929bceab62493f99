/*
 * lk_oracle.h -- CPU oracle for the dense pyramidal Lucas-Kanade hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the algorithm in
 * the reference's SerialPyr.cpp (and of the OpenCV 2.4 imgproc/core semantics
 * it calls).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may link or load it; the product library (liblkflow.so) never does.
 *
 * Parity is PINNED: tests/test_oracle_golden.py checks this file against the
 * reference's own golden fields U2Test/V2Test (single level), UPyrTest/VPyrTest
 * (7-level pyramid) and V2.txt (LK on the in-place-warped inputs).
 *
 * All planes are fp64, row-major, contiguous (pitch == cols).
 */
#ifndef LK_ORACLE_H
#define LK_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* SerialPyr.cpp:196-215 getMaxLayer */
int lko_max_layer(int cols, int rows);

/* SerialPyr.cpp:31-71 get_fx / get_fy / get_ft (2x2 filter2D, anchor (1,1), reflect-101) */
void lko_get_fx(const double *i1, const double *i2, int rows, int cols, double *fx);
void lko_get_fy(const double *i1, const double *i2, int rows, int cols, double *fy);
void lko_get_ft(const double *i1, const double *i2, int rows, int cols, double *ft);

/* SerialPyr.cpp:80-106 get_Sum9 / get_Sum9_Mat, window generalised to win x win (win odd) */
void lko_sum_win(const double *m, int rows, int cols, int win, double *dst);

/* SerialPyr.cpp:121-144 getLucasKanadeOpticalFlow */
void lko_lucas_kanade(const double *i1, const double *i2, int rows, int cols, int win,
                      double *u, double *v);

/* cv::pyrDown (OpenCV 2.4 pyramids.cpp pyrDown_<FltCast<double,8>>), dst is ((rows+1)/2) x ((cols+1)/2) */
void lko_pyr_down(const double *src, int rows, int cols, double *dst);

/* cv::pyrUp (OpenCV 2.4 pyramids.cpp pyrUp_<FltCast<double,6>>), dst is (2*rows) x (2*cols) */
void lko_pyr_up(const double *src, int rows, int cols, double *dst);

/* SerialPyr.cpp:179-187 map build; map is interleaved (x,y) float32, sign = +1 (map1) or -1 (map2) */
void lko_build_map(const double *upu, const double *upv, int rows, int cols, float sign, float *map_xy);

/* cv::remap(src, dst, map, Mat(), INTER_LINEAR), BORDER_CONSTANT 0 (OpenCV 2.4 imgwarp.cpp) */
void lko_remap_linear(const double *src, int rows, int cols, const float *map_xy, double *dst);

/* SerialPyr.cpp:157-194 coarseToFineEstimation.  img1/img2 ARE MUTATED in place when
 * levels >= 2, exactly as the reference mutates pyr[0] (header share + copyTo). Returns 0 or
 * a negative error (bad levels / win). */
int lko_coarse_to_fine(double *img1, double *img2, int rows, int cols, int levels, int win,
                       double *u, double *v);

/* SerialPyr.cpp:237 convertTo(CV_64FC1, 1.0/255) */
void lko_u8_to_f64(const unsigned char *src, long n, double *dst);

/* ParallelVideoPyr.cpp:845-873 flow consumer: count / sum x / sum y of pixels with |flow| >= thresh */
void lko_flow_centroid(const double *u, const double *v, int rows, int cols, double thresh,
                       long long *count, double *sum_x, double *sum_y);

/* ParallelVideoPyr.cpp:803-813 -- the step that PRODUCES the hot call's CV_64F inputs from the 8-bit
 * frame difference: threshold(diff, 10, 255, THRESH_TOZERO); Sobel x + Sobel y (CV_64F, 3x3, reflect-101);
 * dilate x2; erode x2 (3x3 rect, outside pixels ignored).  All values are exact integers. */
void lko_edge_map_u8(const unsigned char *diff, int rows, int cols, int thresh, double *out);

/* ParallelVideoPyr.cpp:786-787 GaussianBlur(Size(k,k), sigma) on 8UC3 + cvtColor(BGR2GRAY) -> 8UC1.
 * PARITY UNPINNED (no reference fixture covers the pre-processing chain; OpenCV 2.4 fixed-point restated). */
void lko_gaussian_kernel_q8(int n, double sigma, int *k);
void lko_blur_gray_bgr8(const unsigned char *bgr, int rows, int cols, int ksize, double sigma,
                        unsigned char *gray);

/* ParallelVideoPyr.cpp:800 diff = cur - rate*prev on 8UC1 (addWeighted8u float arithmetic). PARITY UNPINNED. */
void lko_frame_diff_u8(const unsigned char *cur, const unsigned char *prev, long n, double rate,
                       unsigned char *diff);

/* ParallelVideoPyr.cpp:778 resize(frame, .., Size(dcols, drows), 0, 0, INTER_CUBIC) on 8UC3 (OpenCV 2.4 fixed-point
 * tables, SSE2 float vertical pass). PARITY UNPINNED.  _tables: source offset and the four 11-bit weights per index. */
void lko_resize_cubic_tables(int ssize, int dsize, int *ofs, short *coef);
void lko_resize_cubic_bgr8(const unsigned char *src, int srows, int scols, int drows, int dcols, unsigned char *dst);

#ifdef __cplusplus
}
#endif
#endif
