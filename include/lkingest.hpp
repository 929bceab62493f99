// lkingest.hpp -- lkflow::VideoFile: the subset of cv::VideoCapture the reference's video loop uses (ParallelVideoPyr.cpp:769-776:
// construct / open, isOpened, `cap >> frame`) over the C ABI of include/lkingest.h, for hosts whose OpenCV has no video I/O (this
// image has no video decoder at all).  Reads ISO-BMFF files with one H.264 Constrained-Baseline track -- the reference's pacman.mp4.
// Link with -llkingest (optical-flow_amd/liblkingest.so, host-only, built by g++).
//
//     lkflow::VideoFile cap("pacman.mp4");
//     std::vector<unsigned char> gray(cap.rows() * cap.cols());
//     while (cap.readGray(gray.data(), cap.cols())) { ... lkflow::StreamEngine / lkflow_compute_u8 ... }
//
// With the OpenCV headers visible (include after <opencv2/core/core.hpp>, like lkflow.hpp) `cap >> frame` fills a CV_8UC3 cv::Mat.
#ifndef LKINGEST_HPP
#define LKINGEST_HPP

#include <stdexcept>
#include <string>
#include <vector>

#include "lkingest.h"

namespace lkflow {

class VideoFile {
  public:
    VideoFile() = default;
    explicit VideoFile(const std::string &path) { open(path); }
    VideoFile(const VideoFile &) = delete;
    VideoFile &operator=(const VideoFile &) = delete;
    ~VideoFile() { release(); }

    // cv::VideoCapture::open: false when the file cannot be read or holds a coding tool the reader does not implement
    bool open(const std::string &path)
    {
        release();
        status_ = lkingest_open(&h_, path.c_str());
        if (status_ != LKINGEST_OK) { h_ = nullptr; return false; }
        lkingest_info(h_, &rows_, &cols_, &frames_);
        return true;
    }
    bool isOpened() const { return h_ != nullptr; }
    void release() { if (h_) lkingest_close(h_); h_ = nullptr; rows_ = cols_ = frames_ = 0; }
    int rows() const { return rows_; }          // CV_CAP_PROP_FRAME_HEIGHT
    int cols() const { return cols_; }          // CV_CAP_PROP_FRAME_WIDTH
    int frameCount() const { return frames_; }  // CV_CAP_PROP_FRAME_COUNT
    int status() const { return status_; }      // the last LKINGEST_* code
    std::string lastError() const { return h_ ? lkingest_last_error(h_) : std::string(); }

    // the next frame's luma plane (what BGR2GRAY of the decoded frame is a linear map of); false at the end of the stream;
    // throws std::runtime_error when the bit stream does not decode
    bool readGray(unsigned char *y, size_t step_bytes) { return check(lkingest_read_yuv(need(), y, step_bytes, nullptr, nullptr, 0)); }
    // the next frame as interleaved BGR, rows() x cols() x 3 (the layout of `cap >> frame`)
    bool readBGR(unsigned char *bgr, size_t step_bytes) { return check(lkingest_read_bgr(need(), bgr, step_bytes)); }
    bool read(std::vector<unsigned char> &bgr)
    {
        bgr.resize((size_t)rows_ * cols_ * 3);
        return readBGR(bgr.data(), (size_t)cols_ * 3);
    }
#ifdef LKFLOW_HAVE_OPENCV
    bool read(cv::Mat &frame)
    {
        if (!h_) return false;
        frame.create(rows_, cols_, CV_8UC3);
        if (readBGR(frame.data, frame.step)) return true;
        frame.release();                                     // cv::VideoCapture leaves an empty Mat at the end of the stream
        return false;
    }
    VideoFile &operator>>(cv::Mat &frame) { read(frame); return *this; }
#endif

  private:
    lkingest *need() const
    {
        if (!h_) throw std::runtime_error("lkflow::VideoFile: not opened");
        return h_;
    }
    bool check(int rc)
    {
        status_ = rc < 0 ? rc : LKINGEST_OK;
        if (rc < 0) throw std::runtime_error("lkflow::VideoFile: " + lastError());
        return rc == 1;
    }
    lkingest *h_ = nullptr;
    int rows_ = 0, cols_ = 0, frames_ = 0, status_ = LKINGEST_OK;
};

}  // namespace lkflow
#endif
