// TEST INFRASTRUCTURE ONLY -- a 60-line stand-in for <opencv2/core/core.hpp> (OpenCV 2.4) with just the members of
// cv::Mat that include/lkflow.hpp's LKFLOW_HAVE_OPENCV branch and the reference's call sites touch
// (ParallelVideoPyr.cpp:822-832: Mat(), Mat(rows, cols, type), create, zeros, rows, cols, step, data, type(), empty(),
// at<T>, ptr<T>, ROI-style padded steps).  OpenCV is not installed in this image; this header lets the compiler see
// that branch (tests/test_cpp_dropin.py).  It is NOT OpenCV and must never be on a product include path.
#ifndef FAKE_OPENCV_CORE_HPP
#define FAKE_OPENCV_CORE_HPP
#include <cstddef>
#include <cstring>
#include <memory>

#define CV_64F 6
#define CV_64FC1 6
#define CV_8UC1 0

namespace cv {
class Mat {
public:
    struct Step {                                  // cv::Mat::step is a MatStep that converts to size_t
        size_t v = 0;
        operator size_t() const { return v; }
        Step &operator=(size_t s) { v = s; return *this; }
    };
    int rows = 0, cols = 0;
    Step step;
    unsigned char *data = nullptr;

    Mat() = default;
    Mat(int r, int c, int t) { create(r, c, t); }
    Mat(int r, int c, int t, size_t step_bytes) { type_ = t; alloc(r, c, step_bytes); }   // (test helper: padded rows, like a ROI)
    static Mat zeros(int r, int c, int t) { return Mat(r, c, t); }
    void create(int r, int c, int t)
    {
        if (data && r == rows && c == cols && t == type_) return;
        type_ = t;
        alloc(r, c, (size_t)c * elem());
    }
    int type() const { return type_; }
    bool empty() const { return !data || rows == 0 || cols == 0; }
    template <class T> T *ptr(int y = 0) { return reinterpret_cast<T *>(data + (size_t)y * step.v); }
    template <class T> const T *ptr(int y = 0) const { return reinterpret_cast<const T *>(data + (size_t)y * step.v); }
    template <class T> T &at(int y, int x) { return ptr<T>(y)[x]; }
    template <class T> const T &at(int y, int x) const { return ptr<T>(y)[x]; }

private:
    size_t elem() const { return type_ == CV_64FC1 ? 8 : 1; }
    void alloc(int r, int c, size_t st)
    {
        rows = r; cols = c; step = st;
        buf_.reset(new unsigned char[(size_t)r * st + 8]());      // shared like cv::Mat's ref-counted header
        data = buf_.get();
    }
    int type_ = CV_64FC1;
    std::shared_ptr<unsigned char[]> buf_;
};
}  // namespace cv
#endif
