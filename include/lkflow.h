/*
 * lkflow.h -- C ABI of the MI355X dense pyramidal Lucas-Kanade engine (liblkflow.so).
 *
 * Drop-in boundary for the ONE hot call of mileslindheimer/optical-flow:
 *
 *     void coarseToFineEstimation(Mat &img1, Mat &img2, Mat &u, Mat &v, int nLevels)
 *         ParallelVideoPyr.cpp:578 (OpenMP)   SerialPyr.cpp:157 (serial ground truth)
 *         called from the video loop at ParallelVideoPyr.cpp:832 and the still-image
 *         harness at ParallelBasicPyr.cpp:753
 *     void getLucasKanadeOpticalFlow(Mat &img1, Mat &img2, Mat &u, Mat &v)
 *         ParallelVideoPyr.cpp:327   SerialPyr.cpp:121
 *     int  getMaxLayer(Mat &img)      ParallelVideoPyr.cpp:545   SerialPyr.cpp:196
 *
 * The reference has no FFI; these entry points are what a binding for that call would bind
 * (plain pointers and sizes, no C++ / torch / OpenCV types).  include/lkflow.hpp wraps them in
 * the reference's own C++ signatures (cv::Mat-compatible) -- see INTEGRATION.md.
 *
 * All image planes are CV_64FC1: fp64, row-major, "step" = bytes between rows (>= cols*8,
 * multiple of 8).  Every function returns LKFLOW_OK (0) or a negative lkflow_status.
 * There is NO CPU fallback: without a HIP device every compute entry point fails with
 * LKFLOW_E_NODEVICE / LKFLOW_E_HIP.
 */
#ifndef LKFLOW_H
#define LKFLOW_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum lkflow_status {
    LKFLOW_OK = 0,
    LKFLOW_E_ARG = -1,      /* null pointer, rows/cols < 1, bad step, n_pairs < 1, bad flags      */
    LKFLOW_E_LEVELS = -2,   /* pyr_levels < 1 or > lkflow_max_levels(cols, rows)                   */
                            /*   (the reference throws cv::Exception at "tmpu += upu", :635)       */
    LKFLOW_E_WIN = -3,      /* win even, < 3 or > LKFLOW_MAX_WIN                                   */
    LKFLOW_E_HIP = -4,      /* a HIP runtime call failed (see lkflow_last_error)                   */
    LKFLOW_E_NOMEM = -5,    /* device workspace allocation failed                                  */
    LKFLOW_E_NODEVICE = -6  /* no HIP device / bad device index                                    */
} lkflow_status;

#define LKFLOW_MAX_WIN 31

/* flags */
#define LKFLOW_MUTATE_INPUTS 1u /* write the level-0 warped images back into prev/next exactly as
                                   the reference overwrites pyr[0] (ParallelVideoPyr.cpp:568,696,701).
                                   Default (0): inputs are left untouched, so frame pairs are
                                   independent and can be batched / sharded.  Only with n_pairs==1. */

typedef struct lkflow_ctx lkflow_ctx;

/* getMaxLayer (ParallelVideoPyr.cpp:545-564): the largest legal pyr_levels for cols x rows. */
int lkflow_max_levels(int cols, int rows);

const char *lkflow_strerror(int status);
/* Text of the last HIP error seen by this context ("" if none). */
const char *lkflow_last_error(const lkflow_ctx *ctx);
/* Library version string. */
const char *lkflow_version(void);
/* Diagnostics / measurement: the name (as rocprofv3 prints it) of the LK kernel the library launches for one pyramid level of
 * rows x cols with window `win` in a batch of n_pairs; has_up = the level is below the coarsest (it adds the up-sampled coarse
 * flow), fused_ok = the call does not need the level's warped images in memory (i.e. not level 0 of a LKFLOW_MUTATE_INPUTS
 * call).  The answer is for a context with DEFAULT options on device 0 (LKFLOW_OPT_IMPL, LKFLOW_OPT_FUSE_WARP and
 * LKFLOW_OPT_LK_BATCH select other kernels); the shipped library has no other knob that could change it.  bench.py labels its
 * roofline with it. */
int lkflow_lk_kernel_name(int rows, int cols, int win, int has_up, int fused_ok, int n_pairs, char *buf, size_t n);
/* Diagnostics: how the wave-specialised LK kernel's launch partitions `column_strips` (column blocks x pairs) of `rows` rows
 * over `slots` resident workgroups when a strip marches 2 * halo_rows + 6 rows more than it stores (lone_slots: see
 * DESIGN.md section 4.1): out = {rows per strip of the first phase, column strips in the first phase, rows per strip of the
 * second phase}; one height <=> out[1] == column_strips.  Pure host arithmetic (no GPU): tests/test_abi.py walks the
 * workgroup -> tile map of such plans. */
int lkflow_strip_plan(int rows, int halo_rows, long long column_strips, int slots, int lone_slots, long long out[3]);

/* One context per GPU and caller thread: owns the device workspace (pyramids, per-level flow). */
int lkflow_ctx_create(lkflow_ctx **out, int device);
void lkflow_ctx_destroy(lkflow_ctx *ctx);
/* Optional: size the workspace up front (needed before HIP-graph capture; otherwise it grows
 * lazily with a stream-synchronising reallocation). */
int lkflow_ctx_reserve(lkflow_ctx *ctx, int rows, int cols, int pyr_levels, int n_pairs);
/* Tuning / debug knobs (never needed for correctness). */
#define LKFLOW_OPT_IMPL 1      /* 0 = row-marching LK kernel (default), 1 = per-pixel cross-check LK kernel */
#define LKFLOW_OPT_TILE_ROWS 2 /* output rows per LK block; 0 = automatic */
#define LKFLOW_OPT_PROFILE 3   /* 0 = off (default); -1 = bracket EVERY kernel launch with HIP events on
                                  its own stream; 1 + stage*32 + level = only that (stage, level).
                                  Read back with lkflow_ctx_profile_read. */
#define LKFLOW_OPT_PIN_HOST 4  /* 0 (default) / 1: page-lock the caller's host buffers in place (hipHostRegister, cached in
                                  the context, released by value 0 or lkflow_ctx_destroy) so that the host-buffer entry
                                  points move data by asynchronous DMA instead of the pageable path.  ONLY for buffers that
                                  outlive the context or are re-used call after call (a video loop's Mats): a registered
                                  buffer must not be freed while the option is on. */
#define LKFLOW_OPT_FUSE_WARP 5 /* windows 5..31 on even levels: the warp of a level runs INSIDE its LK kernel (four "W" waves of the
                                  wave-specialised kernel make the warped rows and hand them over through LDS), so the warped planes
                                  never reach memory; bit-identical to warp + LK.  0 = never, 1 = wherever the kernel takes the
                                  shape, 2 (default) = where it was measured faster: levels of >= 256 rows, and any level of a
                                  launch of at most 1.25 Mi pixels (pairs x rows x cols).  Never at level 0
                                  of a LKFLOW_MUTATE_INPUTS call (the caller wants the warped images). */
#define LKFLOW_OPT_LK_BATCH 6  /* 0 (default) / 1: windows 5..15 on even levels run the batch-synchronous LK kernel (lk_batch.h:
                                  8-wave workgroups, 8 rows per barrier pair, 5 window-sum waves next to 3 producer waves)
                                  instead of the wave-specialised one.  Same expressions; the vertical window sums are
                                  associated in chunks of win + 1 instead of win rows (agreement ~1e-15 relative).  Experimental:
                                  measured slower (round-3 record in DESIGN.md).  Only in a -DLKFLOW_EXPERIMENTAL build of the
                                  library (tools/build_variant.sh); the shipped one answers LKFLOW_E_ARG to value 1. */
int lkflow_ctx_set_option(lkflow_ctx *ctx, int key, int value);

/* Stage ids of the kernels the pyramid driver launches (level = pyramid level the launch works at;
 * for PYRDOWN the destination level). */
#define LKFLOW_STAGE_PYRDOWN 0
#define LKFLOW_STAGE_WARP 1
#define LKFLOW_STAGE_LK 2
#define LKFLOW_N_STAGES 3
#define LKFLOW_MAX_LEVELS 32
/* Waits for the recorded events, then fills total_ms[stage*32 + level] / n_launches[stage*32 + level]
 * (arrays of LKFLOW_N_STAGES*LKFLOW_MAX_LEVELS entries) and clears the records. */
int lkflow_ctx_profile_read(lkflow_ctx *ctx, double *total_ms, int *n_launches);
/* Bytes of device workspace currently held. */
size_t lkflow_ctx_workspace_bytes(const lkflow_ctx *ctx);

/*
 * coarseToFineEstimation on HOST buffers (the cv::Mat drop-in, ParallelVideoPyr.cpp:578):
 * copies prev/next to the GPU, runs the pyramid, copies U/V back, synchronises.
 * prev/next are only written when flags has LKFLOW_MUTATE_INPUTS.
 */
int lkflow_compute_f64(lkflow_ctx *ctx, double *prev, double *next, int rows, int cols,
                       size_t step_bytes, double *U, double *V, size_t uv_step_bytes,
                       int pyr_levels, int win, unsigned flags);

/*
 * The same call for 8-bit frames (what SerialPyr.cpp:223-238 loads before convertTo(CV_64FC1, 1.0/255)): the
 * conversion runs on the device (exactly double(p) * (1.0/255)), so 1 byte per pixel crosses PCIe instead of 8.
 * Inputs are never written (no LKFLOW_MUTATE_INPUTS: a CV_8U frame cannot hold the warped CV_64F image).
 */
int lkflow_compute_u8(lkflow_ctx *ctx, const unsigned char *prev, const unsigned char *next, int rows, int cols,
                      size_t step_bytes, double *U, double *V, size_t uv_step_bytes, int pyr_levels, int win);

/*
 * Asynchronous forms of the two host-buffer calls: everything -- upload, pyramid, download -- is
 * enqueued on the context's own stream and the call returns; lkflow_wait() blocks until U, V (and, with
 * LKFLOW_MUTATE_INPUTS, prev/next) are complete.  The host buffers must stay valid and untouched until then.
 * One pair per context may be in flight (a second enqueue first waits for the first); use two contexts to overlap
 * the transfers of one pair with the kernels of another.  With pageable host memory the runtime's copies are
 * staged and partly synchronous; with LKFLOW_OPT_PIN_HOST (or hipHostMalloc'ed buffers) they are true async DMA.
 * The pair in flight works in the context's workspace: a device-batch call (lkflow_compute_batch_f64_dev) on the same
 * context and another stream first waits for it (as lkflow_lk_f64 does).  If an enqueue fails half way, the stream is
 * drained before the call returns, so the caller may free its buffers on any error.  LKFLOW_OPT_PIN_HOST registrations
 * belong to the context that made them: switch the option off (or destroy that context) only while every context that
 * uses the same host buffers is idle.
 */
int lkflow_enqueue_pair_f64(lkflow_ctx *ctx, double *prev, double *next, int rows, int cols, size_t step_bytes,
                            double *U, double *V, size_t uv_step_bytes, int pyr_levels, int win, unsigned flags);
int lkflow_enqueue_pair_u8(lkflow_ctx *ctx, const unsigned char *prev, const unsigned char *next, int rows, int cols,
                           size_t step_bytes, double *U, double *V, size_t uv_step_bytes, int pyr_levels, int win);
int lkflow_wait(lkflow_ctx *ctx);

/* getLucasKanadeOpticalFlow on HOST buffers (ParallelVideoPyr.cpp:327), window generalised. */
int lkflow_lk_f64(lkflow_ctx *ctx, const double *img1, const double *img2, int rows, int cols,
                  size_t step_bytes, double *U, double *V, size_t uv_step_bytes, int win);

/*
 * Device-resident, batched, asynchronous form (what the video loop / bench uses):
 * pair p reads  prev + p*prev_pair_stride_bytes , next + p*next_pair_stride_bytes
 *        writes U + p*uv_pair_stride_bytes , V + p*uv_pair_stride_bytes.
 * A video of F frames with frame stride S is F-1 pairs with prev = frames, next = frames + S,
 * both pair strides = S.  All pointers are DEVICE pointers on the context's GPU.  Work is
 * enqueued on `stream` (a hipStream_t, NULL = default stream); the call does not synchronise and, after
 * lkflow_ctx_reserve() for the shape, makes no allocation: it may be captured into a HIP graph and replayed
 * (tests/test_gpu_round3.py::test_graph_capture_after_reserve_replays_bitwise).  One exception: while a host-buffer pair
 * enqueued on this context (lkflow_enqueue_pair_*) has not been waited for, a call on another stream first synchronises with it
 * on the host -- and is refused with LKFLOW_E_ARG if that stream is being captured: call lkflow_wait() before capturing.
 */
int lkflow_compute_batch_f64_dev(lkflow_ctx *ctx, double *prev, double *next,
                                 size_t prev_pair_stride_bytes, size_t next_pair_stride_bytes,
                                 int n_pairs, int rows, int cols, size_t step_bytes, double *U,
                                 double *V, size_t uv_pair_stride_bytes, size_t uv_step_bytes,
                                 int pyr_levels, int win, unsigned flags, void *stream);

/* Single-level LK, device-resident, batched (same addressing rules). */
int lkflow_lk_batch_f64_dev(lkflow_ctx *ctx, const double *img1, const double *img2,
                            size_t img1_pair_stride_bytes, size_t img2_pair_stride_bytes,
                            int n_pairs, int rows, int cols, size_t step_bytes, double *U, double *V,
                            size_t uv_pair_stride_bytes, size_t uv_step_bytes, int win, void *stream);

/* The same with the accumulate step of the pyramid loop (ParallelVideoPyr.cpp:634-637): U = LK(img1, img2) + pyrUp(cu),
 * V = LK(..) + pyrUp(cv) for a coarse flow (cu, cv) of (rows/2) x (cols/2); rows and cols even.  What every level below
 * the coarsest runs after its warp -- exposed so that the "+= pyrUp" path of the LK kernels can be parity-tested in
 * isolation (non-finite coarse flow included). */
int lkflow_lk_up_batch_f64_dev(lkflow_ctx *ctx, const double *img1, const double *img2,
                               size_t img1_pair_stride_bytes, size_t img2_pair_stride_bytes,
                               int n_pairs, int rows, int cols, size_t step_bytes, const double *cu, const double *cv,
                               size_t c_pair_stride_bytes, size_t c_step_bytes, double *U, double *V,
                               size_t uv_pair_stride_bytes, size_t uv_step_bytes, int win, void *stream);

/* ---- stage entry points (device pointers, n_planes contiguous-or-strided planes) ------------
 * Exposed so that every kernel can be parity-tested against the oracle in isolation. */

/* cv::pyrDown (ParallelVideoPyr.cpp:571): dst is ((rows+1)/2) x ((cols+1)/2). */
int lkflow_pyrdown_f64_dev(lkflow_ctx *ctx, const double *src, size_t src_plane_stride_bytes,
                           int n_planes, int rows, int cols, size_t src_step_bytes, double *dst,
                           size_t dst_plane_stride_bytes, size_t dst_step_bytes, void *stream);

/* Two levels of getGaussianPyramid (ParallelVideoPyr.cpp:566-575) in one pass: dst1 = pyrDown(src) (rows/2 x cols/2), dst2 =
 * pyrDown(dst1) (rows/4 x cols/4), bit-identical to two lkflow_pyrdown_f64_dev calls; the source is read once and dst1 is not
 * read back.  rows and cols multiples of 4, cols >= 16, rows >= 8 (other shapes: LKFLOW_E_ARG -- the pyramid driver then runs
 * two single-level passes). */
int lkflow_pyrdown2_f64_dev(lkflow_ctx *ctx, const double *src, size_t src_plane_stride_bytes, int n_planes, int rows,
                            int cols, size_t src_step_bytes, double *dst1, size_t dst1_plane_stride_bytes,
                            size_t dst1_step_bytes, double *dst2, size_t dst2_plane_stride_bytes, size_t dst2_step_bytes,
                            void *stream);

/* cv::pyrUp (ParallelVideoPyr.cpp:646-647): dst is (2*rows) x (2*cols). */
int lkflow_pyrup_f64_dev(lkflow_ctx *ctx, const double *src, size_t src_plane_stride_bytes,
                         int n_planes, int rows, int cols, size_t src_step_bytes, double *dst,
                         size_t dst_plane_stride_bytes, size_t dst_step_bytes, void *stream);

/*
 * pyrUp + map build + remap of both images (ParallelVideoPyr.cpp:646-703):
 * (cu,cv) is the coarse flow, (rows/2) x (cols/2); img1/img2 are rows x cols (both even);
 * out1 = remap(img1, (x,y) + up(c)/2), out2 = remap(img2, (x,y) - up(c)/2).  out may not alias in.
 */
int lkflow_warp_f64_dev(lkflow_ctx *ctx, const double *img1, const double *img2,
                        size_t img_plane_stride_bytes, int n_pairs, int rows, int cols,
                        size_t img_step_bytes, const double *cu, const double *cv,
                        size_t c_plane_stride_bytes, size_t c_step_bytes, double *out1, double *out2,
                        size_t out_plane_stride_bytes, size_t out_step_bytes, void *stream);

/* The 2 x 2 solve + cv::divide (x / 0 -> 0) of one LK level (ParallelVideoPyr.cpp:470-472, 510-520) on n pixels: S holds
 * five contiguous planes of n window sums (Sxx, Syy, Sxy, Sxt, Syt).  plain_divisions = 0: as every LK kernel computes it (both
 * quotients share one reciprocal iteration, each correctly rounded), 1: two independent IEEE divisions -- the cross-check. */
int lkflow_solve_f64_dev(lkflow_ctx *ctx, const double *S, size_t n, double *u, double *v, int plain_divisions,
                         void *stream);

/* convertTo(CV_64FC1, 1.0/255) (SerialPyr.cpp:237) for n contiguous bytes. */
int lkflow_u8_to_f64_dev(lkflow_ctx *ctx, const unsigned char *src, double *dst, size_t n,
                         void *stream);

/*
 * First step of the video loop's frame pre-processing (ParallelVideoPyr.cpp:778): resize(frame, current_frame,
 * Size(dst_cols, dst_rows), 0, 0, INTER_CUBIC) on 8UC3 (BGR, interleaved) frames; OpenCV 2.4's 8-bit path (11-bit
 * fixed-point cubic weights, A = -0.75; clamped borders; x86 SSE2 build: the vertical pass runs in float for all but
 * the last (dst_cols * 3) mod 8 bytes of a row).  The reference calls it with Size(1080, 1080).
 * src: n_frames frames of src_rows x src_cols x 3 bytes (row step src_step_bytes); dst: dst_rows x dst_cols x 3 each.
 */
int lkflow_resize_cubic_bgr8_dev(lkflow_ctx *ctx, const unsigned char *src, size_t src_plane_stride_bytes,
                                 int n_frames, int src_rows, int src_cols, size_t src_step_bytes,
                                 unsigned char *dst, size_t dst_plane_stride_bytes, int dst_rows, int dst_cols,
                                 size_t dst_step_bytes, void *stream);

/*
 * Frame pre-processing of the video loop (ParallelVideoPyr.cpp:786-787): GaussianBlur(frame, frame,
 * Size(ksize,ksize), sigma, sigma) on the 8UC3 (BGR, interleaved) frame followed by
 * cvtColor(.., CV_BGR2GRAY); OpenCV 2.4's 8-bit fixed-point arithmetic (x86 SSE2 build: column pass rounds
 * half to even).  The reference calls it with ksize 9, sigma 1.5.  ksize odd, <= 15; sigma > 0.
 * bgr: n_frames frames of rows x cols x 3 bytes (row step src_step_bytes); gray: rows x cols bytes each.
 * (The resize to 1080 x 1080 that precedes it, :778: lkflow_resize_cubic_bgr8_dev.)
 */
int lkflow_blur_gray_bgr8_dev(lkflow_ctx *ctx, const unsigned char *bgr, size_t src_plane_stride_bytes,
                              int n_frames, int rows, int cols, size_t src_step_bytes, int ksize, double sigma,
                              unsigned char *gray, size_t dst_plane_stride_bytes, size_t dst_step_bytes,
                              void *stream);

/*
 * diff = cur - rate * prev on 8UC1 frames (ParallelVideoPyr.cpp:800, rate = LEARNING_RATE 0.3), i.e.
 * cv::addWeighted(cur, 1, prev, -rate, 0): float arithmetic, round half to even, saturated to 0..255.
 * Plane p reads cur + p*stride and prev + p*stride (a video: prev = gray, cur = gray + frame stride).
 */
int lkflow_frame_diff_u8_dev(lkflow_ctx *ctx, const unsigned char *cur, const unsigned char *prev,
                             size_t src_plane_stride_bytes, int n_planes, int rows, int cols,
                             size_t src_step_bytes, double rate, unsigned char *diff,
                             size_t dst_plane_stride_bytes, size_t dst_step_bytes, void *stream);

/*
 * The video loop's last pre-processing step, i.e. what PRODUCES the hot call's CV_64F inputs from the
 * 8-bit frame difference (ParallelVideoPyr.cpp:803-813): threshold(diff, thresh, 255, THRESH_TOZERO);
 * Sobel x + Sobel y (CV_64F, 3x3, reflect-101); dilate x2; erode x2 (3x3 rectangle, outside ignored).
 * src: n_planes uint8 planes (rows x cols, row step src_step_bytes); dst: n_planes fp64 planes.
 */
int lkflow_edge_map_u8_dev(lkflow_ctx *ctx, const unsigned char *src, size_t src_plane_stride_bytes,
                           int n_planes, int rows, int cols, size_t src_step_bytes, int thresh, double *dst,
                           size_t dst_plane_stride_bytes, size_t dst_step_bytes, void *stream);

/*
 * lkflow_frame_diff_u8_dev + lkflow_edge_map_u8_dev in one pass (ParallelVideoPyr.cpp:800-813): the
 * difference image never reaches memory.  Same results as the two calls in sequence.
 */
int lkflow_diff_edge_map_u8_dev(lkflow_ctx *ctx, const unsigned char *cur, const unsigned char *prev,
                                size_t src_plane_stride_bytes, int n_planes, int rows, int cols,
                                size_t src_step_bytes, double rate, int thresh, double *dst,
                                size_t dst_plane_stride_bytes, size_t dst_step_bytes, void *stream);

/*
 * Flow consumer of the video loop (ParallelVideoPyr.cpp:845-873), batched and fused:
 * for each pair, out[3*p+0] = number of pixels with sqrt(u*u+v*v) >= thresh,
 * out[3*p+1] = sum of their column indices, out[3*p+2] = sum of their row indices (fp64, exact
 * integers), so that only 3 scalars per pair leave the GPU instead of two planes.
 */
int lkflow_flow_centroid_f64_dev(lkflow_ctx *ctx, const double *U, const double *V,
                                 size_t uv_pair_stride_bytes, int n_pairs, int rows, int cols,
                                 size_t uv_step_bytes, double thresh, double *out, void *stream);

/* ---- frame-queue engine over a device list (SURVEY.md 8(b) "lkflow_ctx_create(devices...), lkflow_enqueue_pair,
 * lkflow_wait"; 8(e)) -------------------------------------------------------------------------------------------------
 * The reference's video loop (ParallelVideoPyr.cpp:769-873) for a caller that holds a whole stream of 8-bit frames in ONE
 * process: F frames = F - 1 consecutive frame pairs, partitioned over the listed devices in contiguous blocks with a
 * one-frame overlap (device k of n gets frames [first_k, first_k + n_k): lkflow_stream_block, the rule of SURVEY.md 8(e));
 * one worker thread + context + HIP stream per list entry (a device may be listed more than once).  Per block: uint8
 * frames host -> device, convertTo(CV_64FC1, 1/255) on the device (SerialPyr.cpp:237), coarseToFineEstimation on the
 * block's pairs (inputs not mutated: pairs are independent), the consumer's |flow| >= thresh centroid sums
 * (ParallelVideoPyr.cpp:845-873); 3 doubles per pair come back.  Every pair's result is bit-identical to the pair computed
 * alone on any device, so the output does not depend on the device list.  No collective is involved (one process; the
 * multi-process form over RCCL is optical-flow_amd/sharding.py + bench.py --gpus N). */
typedef struct lkflow_stream lkflow_stream;
#define LKFLOW_STREAM_KEEP_FLOW 1u /* keep every pair's U, V on its device for lkflow_stream_flow (2 planes per pair) */
int lkflow_stream_create(lkflow_stream **out, const int *devices, int n_devices, int rows, int cols, int pyr_levels,
                         int win, unsigned flags);
void lkflow_stream_destroy(lkflow_stream *s);
/* Device memory one worker may use per lkflow_compute_batch_f64_dev call (default 16 GiB): a block is processed in chunks
 * of pairs that fit it.  Results do not depend on it. */
int lkflow_stream_set_budget(lkflow_stream *s, size_t bytes_per_device);
/* Start the flow of a stream: n_frames >= 2 frames of rows x cols bytes in HOST memory, frame f at frames + f *
 * frame_stride_bytes, row step step_bytes.  Returns at once; the frames must stay valid and unchanged until
 * lkflow_stream_centroids / _flow / _destroy has returned.  One push may be in flight: another first completes it. */
int lkflow_stream_push_u8(lkflow_stream *s, const unsigned char *frames, int n_frames, size_t frame_stride_bytes,
                          size_t step_bytes, double thresh);
/* Wait for the push; out[3*p + {0,1,2}] = count, sum of column indices, sum of row indices of pair p's pixels with
 * sqrt(u*u + v*v) >= thresh (lkflow_flow_centroid_f64_dev's numbers).  n_pairs must be n_frames - 1.  A worker's failure
 * is reported here (lkflow_stream_last_error names the device). */
int lkflow_stream_centroids(lkflow_stream *s, double *out, int n_pairs);
/* U, V of one pair of the last push into host buffers (only with LKFLOW_STREAM_KEEP_FLOW). */
int lkflow_stream_flow(lkflow_stream *s, int pair, double *U, double *V, size_t uv_step_bytes);
/* The partition rule: frames [*first_frame, *first_frame + *n_frames_local) for list entry k of n_devices
 * (*n_frames_local == 0: more devices than pairs, the entry idles). */
int lkflow_stream_block(int n_frames, int n_devices, int k, int *first_frame, int *n_frames_local);
const char *lkflow_stream_last_error(const lkflow_stream *s);

#ifdef __cplusplus
}
#endif
#endif /* LKFLOW_H */
