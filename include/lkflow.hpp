// lkflow.hpp -- header-only C++ drop-in over the C ABI (lkflow.h) with the reference's own signatures.
//
//   void coarseToFineEstimation(Mat &img1, Mat &img2, Mat &u, Mat &v, int nLevels);   ParallelVideoPyr.cpp:578
//   void getLucasKanadeOpticalFlow(Mat &img1, Mat &img2, Mat &u, Mat &v);             ParallelVideoPyr.cpp:327
//   int  getMaxLayer(Mat &img);                                                       ParallelVideoPyr.cpp:545
//   void computeFlow(const Mat &frame_prev, const Mat &frame_next, Mat &U, Mat &V, int pyr_levels, int win = 3);
//        (BASELINE.json's name for the same call: window generalised, inputs NOT mutated)
//
// `Mat` is cv::Mat when the OpenCV headers are visible (define nothing, just include this file after
// <opencv2/core/core.hpp>); otherwise lkflow::Mat below, a minimal CV_64FC1 stand-in with the members
// the reference's call sites touch (rows, cols, step, data, at<double>, ptr<double>, zeros, create).
//
// Error behaviour mirrors the reference: where OpenCV would throw cv::Exception (level count larger
// than getMaxLayer -> size mismatch at ":635 tmpu += upu", wrong type, empty image) these wrappers
// throw std::invalid_argument; device failures throw std::runtime_error.  There is no CPU fallback.
#ifndef LKFLOW_HPP
#define LKFLOW_HPP

#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "lkflow.h"

#if defined(__has_include)
#if __has_include(<opencv2/core/core.hpp>) && !defined(LKFLOW_NO_OPENCV)
#include <opencv2/core/core.hpp>
#define LKFLOW_HAVE_OPENCV 1
#endif
#endif

namespace lkflow {

#ifdef LKFLOW_HAVE_OPENCV
using Mat = cv::Mat;
inline bool mat_ok(const Mat &m) { return !m.empty() && m.type() == CV_64FC1; }
inline void mat_create(Mat &m, int rows, int cols) { m.create(rows, cols, CV_64FC1); }
inline double *mat_data(const Mat &m) { return reinterpret_cast<double *>(m.data); }
inline size_t mat_step(const Mat &m) { return m.step; }
#else
// Minimal CV_64FC1 matrix (owning, row-major, step in BYTES like cv::Mat::step).
class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    unsigned char *data = nullptr;

    Mat() = default;
    Mat(int r, int c) { create(r, c); }
    Mat(const Mat &o) { *this = o; }
    Mat(Mat &&o) noexcept { swap(o); }
    ~Mat() { std::free(data); }
    Mat &operator=(const Mat &o)
    {
        if (this != &o) {
            create(o.rows, o.cols);
            for (int y = 0; y < rows; ++y) std::memcpy(ptr<double>(y), o.ptr<double>(y), (size_t)cols * 8);
        }
        return *this;
    }
    Mat &operator=(Mat &&o) noexcept { swap(o); return *this; }
    void swap(Mat &o) noexcept
    {
        std::swap(rows, o.rows); std::swap(cols, o.cols); std::swap(step, o.step); std::swap(data, o.data);
    }
    void create(int r, int c)
    {
        if (r == rows && c == cols && data) return;
        std::free(data);
        rows = r; cols = c; step = (size_t)c * sizeof(double);
        data = static_cast<unsigned char *>(std::calloc((size_t)r * c + 1, sizeof(double)));
        if (!data) throw std::bad_alloc();
    }
    static Mat zeros(int r, int c) { return Mat(r, c); }
    bool empty() const { return !data || rows == 0 || cols == 0; }
    template <class T> T *ptr(int y) { return reinterpret_cast<T *>(data + (size_t)y * step); }
    template <class T> const T *ptr(int y) const { return reinterpret_cast<const T *>(data + (size_t)y * step); }
    template <class T> T &at(int y, int x) { return ptr<T>(y)[x]; }
    template <class T> const T &at(int y, int x) const { return ptr<T>(y)[x]; }
};
inline bool mat_ok(const Mat &m) { return !m.empty(); }
inline void mat_create(Mat &m, int rows, int cols) { m.create(rows, cols); }
inline double *mat_data(const Mat &m) { return reinterpret_cast<double *>(m.data); }
inline size_t mat_step(const Mat &m) { return m.step; }
#endif

// RAII lkflow_ctx; one per thread and GPU.  The free functions below use a thread-local default.
class Engine {
public:
    explicit Engine(int device = 0)
    {
        const int rc = lkflow_ctx_create(&ctx_, device);
        if (rc != LKFLOW_OK) throw std::runtime_error(std::string("lkflow_ctx_create: ") + lkflow_strerror(rc));
    }
    ~Engine() { lkflow_ctx_destroy(ctx_); }
    Engine(const Engine &) = delete;
    Engine &operator=(const Engine &) = delete;
    lkflow_ctx *handle() const { return ctx_; }

    void computeFlow(const Mat &prev, const Mat &next, Mat &U, Mat &V, int pyr_levels, int win = 3,
                     unsigned flags = 0)
    {
        enqueue(prev, next, U, V, pyr_levels, win, flags);
        wait();
    }
    // Asynchronous form: upload, pyramid and download are enqueued on the engine's stream;
    // wait() completes U and V.  prev, next, U, V must stay alive and untouched until then.  One pair per
    // engine may be in flight; two engines overlap the transfers of one pair with the kernels of another.
    void enqueue(const Mat &prev, const Mat &next, Mat &U, Mat &V, int pyr_levels, int win = 3, unsigned flags = 0)
    {
        check_pair(prev, next);
        // A pair still in flight may be copying from / into the engine-owned staging buffer (unequal steps): complete it
        // BEFORE that buffer is read back, rewritten or re-allocated (the C ABI's own "second enqueue waits for the first"
        // happens inside lkflow_enqueue_pair_f64, i.e. too late for staged_).
        drain();
        seat_outputs(U, V, prev.rows, prev.cols);
        double *nx = mat_data(next);
        if (mat_step(next) != mat_step(prev)) {
            // The reference takes any two CV_64FC1 Mats of one size (a ROI next to a continuous matrix); the C ABI describes
            // the pair with ONE row step: `next` goes through an engine-owned buffer with prev's step (and comes back from
            // it after the call when the level-0 warp overwrites the inputs, ParallelVideoPyr.cpp:696,701).
            const size_t st = mat_step(prev), row = (size_t)prev.cols * sizeof(double);
            stage_rows(reinterpret_cast<const unsigned char *>(mat_data(next)), mat_step(next), prev.rows, st, row);
            nx = staged_.data();
            if (flags & LKFLOW_MUTATE_INPUTS) {
                back_dst_ = reinterpret_cast<unsigned char *>(mat_data(next)); back_step_ = mat_step(next);
                back_rows_ = prev.rows; back_row_bytes_ = row; back_src_step_ = st;
            }
        }
        const int rc = lkflow_enqueue_pair_f64(ctx_, mat_data(prev), nx, prev.rows, prev.cols, mat_step(prev),
                                               mat_data(U), mat_data(V), mat_step(U), pyr_levels, win, flags);
        if (rc != LKFLOW_OK) back_dst_ = nullptr;             // (a failed enqueue has drained its stream and written nothing back)
        else inflight_ = true;
        check(rc);
    }
    void wait() { drain(); }
    // Page-lock the (re-used) Mats of a video loop in place: see LKFLOW_OPT_PIN_HOST in lkflow.h.
    void pinHostBuffers(bool on)
    {
        drain();
        check(lkflow_ctx_set_option(ctx_, LKFLOW_OPT_PIN_HOST, on ? 1 : 0));
        pinned_ = on;
    }

    void getLucasKanadeOpticalFlow(const Mat &img1, const Mat &img2, Mat &u, Mat &v, int win = 3)
    {
        check_pair(img1, img2);
        drain();                                             // (as in enqueue(): before staged_ is touched)
        seat_outputs(u, v, img1.rows, img1.cols);
        const double *p2 = mat_data(img2);
        if (mat_step(img2) != mat_step(img1)) {              // (as in enqueue(): one row step per pair in the C ABI)
            stage_rows(reinterpret_cast<const unsigned char *>(mat_data(img2)), mat_step(img2), img1.rows, mat_step(img1),
                       (size_t)img1.cols * sizeof(double));
            p2 = staged_.data();
        }
        check(lkflow_lk_f64(ctx_, mat_data(img1), p2, img1.rows, img1.cols, mat_step(img1),
                            mat_data(u), mat_data(v), mat_step(u), win));
    }

private:
    // The reference re-seats u, v to new H x W buffers (`u = tmpu`, :639-640).  cv::Mat::create keeps a matrix
    // of matching size and type as it is -- including a caller's ROI / padded rows, whose step the C ABI's one
    // uv_step_bytes cannot describe when U and V differ: such outputs are re-seated to fresh continuous
    // matrices (which is what the reference's header assignment amounts to).
    static void seat_outputs(Mat &U, Mat &V, int rows, int cols)
    {
        mat_create(U, rows, cols);
        mat_create(V, rows, cols);
        if (mat_step(U) != mat_step(V) || mat_step(U) < (size_t)cols * sizeof(double) || mat_step(U) % sizeof(double)) {
            U = Mat(); V = Mat();
            mat_create(U, rows, cols);
            mat_create(V, rows, cols);
        }
    }
    static void check_pair(const Mat &a, const Mat &b)
    {
        if (!mat_ok(a) || !mat_ok(b)) throw std::invalid_argument("lkflow: images must be non-empty CV_64FC1");
        if (a.rows != b.rows || a.cols != b.cols)
            throw std::invalid_argument("lkflow: image pair must have equal size");
    }
    void check(int rc) const
    {
        if (rc == LKFLOW_OK) return;
        std::string msg = std::string("lkflow: ") + lkflow_strerror(rc);
        if (rc == LKFLOW_E_HIP || rc == LKFLOW_E_NOMEM || rc == LKFLOW_E_NODEVICE)
            throw std::runtime_error(msg + " [" + lkflow_last_error(ctx_) + "]");
        throw std::invalid_argument(msg);
    }
    // Complete the pair in flight (if any), then hand a staged, overwritten `next` back.  Every path that reads, rewrites
    // or re-allocates staged_ goes through here first.
    void drain()
    {
        if (inflight_) {
            inflight_ = false;
            const int rc = lkflow_wait(ctx_);
            if (rc != LKFLOW_OK) back_dst_ = nullptr;
            check(rc);
        }
        finish_staged();
    }
    // Copy `rows` rows of `row` bytes (source step sstep) into the staging buffer with step st.  The buffer only ever GROWS,
    // and only here, i.e. after drain(): nothing is in flight on it.  With pinHostBuffers(true) the library has page-locked
    // the old allocation in place (LKFLOW_OPT_PIN_HOST registers the ranges it is handed): those registrations are dropped
    // BEFORE the vector frees it, and the option is switched back on for the new one.
    void stage_rows(const unsigned char *src, size_t sstep, int rows, size_t st, size_t row)
    {
        const size_t need = (size_t)rows * st / sizeof(double) + 1;
        if (need > staged_.size()) {
            if (pinned_) check(lkflow_ctx_set_option(ctx_, LKFLOW_OPT_PIN_HOST, 0));
            staged_.assign(need, 0.0);
            if (pinned_) check(lkflow_ctx_set_option(ctx_, LKFLOW_OPT_PIN_HOST, 1));
        }
        for (int y = 0; y < rows; ++y)
            std::memcpy(reinterpret_cast<unsigned char *>(staged_.data()) + (size_t)y * st, src + (size_t)y * sstep, row);
    }
    // copy a staged, overwritten `next` back into the caller's matrix (LKFLOW_MUTATE_INPUTS with unequal steps)
    void finish_staged()
    {
        if (back_dst_) {
            for (int y = 0; y < back_rows_; ++y)
                std::memcpy(back_dst_ + (size_t)y * back_step_,
                            reinterpret_cast<const unsigned char *>(staged_.data()) + (size_t)y * back_src_step_, back_row_bytes_);
            back_dst_ = nullptr;
        }
    }
    lkflow_ctx *ctx_ = nullptr;
    std::vector<double> staged_;
    unsigned char *back_dst_ = nullptr;
    size_t back_step_ = 0, back_src_step_ = 0, back_row_bytes_ = 0;
    int back_rows_ = 0;
    bool inflight_ = false, pinned_ = false;
};

// The frame-queue engine over a device list (lkflow_stream_* in lkflow.h): the reference's video loop
// (ParallelVideoPyr.cpp:769-873) for a whole stream of 8-bit frames, frame pairs partitioned over the GPUs of one node inside
// ONE process.  push() returns at once; centroids() waits and returns {count, sum x, sum y} per pair (the consumer's sums,
// :845-873; gestureCentroid() below turns a row into the circle the reference draws).
class StreamEngine {
public:
    StreamEngine(const std::vector<int> &devices, int rows, int cols, int pyr_levels, int win = 3, bool keep_flow = false)
        : rows_(rows), cols_(cols)
    {
        const int rc = lkflow_stream_create(&s_, devices.data(), (int)devices.size(), rows, cols, pyr_levels, win,
                                            keep_flow ? LKFLOW_STREAM_KEEP_FLOW : 0u);
        if (rc == LKFLOW_E_HIP || rc == LKFLOW_E_NOMEM || rc == LKFLOW_E_NODEVICE)
            throw std::runtime_error(std::string("lkflow_stream_create: ") + lkflow_strerror(rc));
        if (rc != LKFLOW_OK) throw std::invalid_argument(std::string("lkflow_stream_create: ") + lkflow_strerror(rc));
    }
    ~StreamEngine() { lkflow_stream_destroy(s_); }
    StreamEngine(const StreamEngine &) = delete;
    StreamEngine &operator=(const StreamEngine &) = delete;
    lkflow_stream *handle() const { return s_; }

    // n_frames contiguous 8-bit frames (rows x cols each, row step = cols unless given); they must outlive the push
    void push(const unsigned char *frames, int n_frames, double thresh = 20.0, size_t frame_stride_bytes = 0, size_t step_bytes = 0)
    {
        const size_t st = step_bytes ? step_bytes : (size_t)cols_, fs = frame_stride_bytes ? frame_stride_bytes : st * (size_t)rows_;
        check(lkflow_stream_push_u8(s_, frames, n_frames, fs, st, thresh));
        n_pairs_ = n_frames - 1;
    }
    std::vector<double> centroids()                          // [n_pairs][3], row-major
    {
        std::vector<double> out((size_t)n_pairs_ * 3);
        check(lkflow_stream_centroids(s_, out.data(), n_pairs_));
        return out;
    }
    void flow(int pair, Mat &U, Mat &V)                      // needs keep_flow
    {
        mat_create(U, rows_, cols_); mat_create(V, rows_, cols_);
        if (mat_step(U) != mat_step(V)) { U = Mat(); V = Mat(); mat_create(U, rows_, cols_); mat_create(V, rows_, cols_); }
        check(lkflow_stream_flow(s_, pair, mat_data(U), mat_data(V), mat_step(U)));
    }
    int pairs() const { return n_pairs_; }

private:
    void check(int rc) const
    {
        if (rc == LKFLOW_OK) return;
        const std::string msg = std::string("lkflow stream: ") + lkflow_strerror(rc) + " [" + lkflow_stream_last_error(s_) + "]";
        if (rc == LKFLOW_E_HIP || rc == LKFLOW_E_NOMEM || rc == LKFLOW_E_NODEVICE) throw std::runtime_error(msg);
        throw std::invalid_argument(msg);
    }
    lkflow_stream *s_ = nullptr;
    int rows_, cols_, n_pairs_ = 0;
};

inline Engine &default_engine()
{
    static thread_local Engine e(0);
    return e;
}

// ---- the reference's free functions ---------------------------------------------------------------
inline int getMaxLayer(const Mat &img) { return lkflow_max_levels(img.cols, img.rows); }

inline void computeFlow(const Mat &frame_prev, const Mat &frame_next, Mat &U, Mat &V, int pyr_levels, int win = 3)
{
    default_engine().computeFlow(frame_prev, frame_next, U, V, pyr_levels, win, 0);
}

// Source-compatible with ParallelVideoPyr.cpp:578 -- INCLUDING the in-place overwrite of img1/img2 by the
// level-0 warp (pyr[0] shares the caller's buffer, :568, :696, :701).
inline void coarseToFineEstimation(Mat &img1, Mat &img2, Mat &u, Mat &v, int nLevels)
{
    default_engine().computeFlow(img1, img2, u, v, nLevels, 3, LKFLOW_MUTATE_INPUTS);
}

inline void getLucasKanadeOpticalFlow(Mat &img1, Mat &img2, Mat &u, Mat &v)
{
    default_engine().getLucasKanadeOpticalFlow(img1, img2, u, v, 3);
}

// ---- host tail of the flow consumer (ParallelVideoPyr.cpp:875-888) ------------------------------------------
// lkflow_flow_centroid_f64_dev returns, per pair, counts = #{|flow| >= thresh} and the sums of their column / row
// indices (:845-873).  What the reference then does with them on the host: avgX /= counts, avgY /= counts (:875-876),
// scale by the INTEGER ratio frame.cols / current_frame.cols (:878-883 -- the display frame over the working
// frame, truncated as the reference's int division truncates) and draw the circle only if counts > 500 (:886).
struct Gesture {
    bool detected;      // counts > 500
    double x, y;        // circle centre in display-frame pixels (valid when detected; 0, 0 for counts == 0)
};
inline Gesture gestureCentroid(double counts, double sum_x, double sum_y, int frame_cols, int work_cols,
                               double min_counts = 500.0)
{
    Gesture g{counts > min_counts, 0.0, 0.0};
    if (counts > 0.0) {
        const int scale = work_cols > 0 ? frame_cols / work_cols : 1;     // int / int, as the reference
        g.x = sum_x / counts * scale;
        g.y = sum_y / counts * scale;
    }
    return g;
}

}  // namespace lkflow
#endif  // LKFLOW_HPP
