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
        check_pair(prev, next);
        mat_create(U, prev.rows, prev.cols);     // the reference re-seats u,v to new H x W buffers (:639-640)
        mat_create(V, prev.rows, prev.cols);
        check(lkflow_compute_f64(ctx_, mat_data(prev), mat_data(next), prev.rows, prev.cols, mat_step(prev),
                                 mat_data(U), mat_data(V), mat_step(U), pyr_levels, win, flags));
    }
    void getLucasKanadeOpticalFlow(const Mat &img1, const Mat &img2, Mat &u, Mat &v, int win = 3)
    {
        check_pair(img1, img2);
        mat_create(u, img1.rows, img1.cols);
        mat_create(v, img1.rows, img1.cols);
        check(lkflow_lk_f64(ctx_, mat_data(img1), mat_data(img2), img1.rows, img1.cols, mat_step(img1),
                            mat_data(u), mat_data(v), mat_step(u), win));
    }

private:
    static void check_pair(const Mat &a, const Mat &b)
    {
        if (!mat_ok(a) || !mat_ok(b)) throw std::invalid_argument("lkflow: images must be non-empty CV_64FC1");
        if (a.rows != b.rows || a.cols != b.cols || mat_step(a) != mat_step(b))
            throw std::invalid_argument("lkflow: image pair must have equal size and step");
    }
    void check(int rc) const
    {
        if (rc == LKFLOW_OK) return;
        std::string msg = std::string("lkflow: ") + lkflow_strerror(rc);
        if (rc == LKFLOW_E_HIP || rc == LKFLOW_E_NOMEM || rc == LKFLOW_E_NODEVICE)
            throw std::runtime_error(msg + " [" + lkflow_last_error(ctx_) + "]");
        throw std::invalid_argument(msg);
    }
    lkflow_ctx *ctx_ = nullptr;
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

}  // namespace lkflow
#endif  // LKFLOW_HPP
