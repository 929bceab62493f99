// opencv_branch_check.cpp -- include/lkflow.hpp's `Mat = cv::Mat` branch (LKFLOW_HAVE_OPENCV) through a compiler and,
// on a GPU box, through one call: the reference's call site (ParallelVideoPyr.cpp:822-832) with cv::Mat arguments whose
// row steps DIFFER (a padded / ROI matrix next to a continuous one -- the reference accepts any two CV_64FC1 Mats).
// Built against tests/cpp/fake_opencv (test infrastructure; OpenCV itself is not installed here).
//   usage: opencv_branch_check            -> compile / link check only (exit 2)
//          opencv_branch_check run        -> needs a GPU
#include <cmath>
#include <cstdio>
#include <cstring>
#include <type_traits>

#include <opencv2/core/core.hpp>
#include "lkflow.hpp"

static_assert(std::is_same<lkflow::Mat, cv::Mat>::value, "with the OpenCV headers visible lkflow::Mat is cv::Mat");
#ifndef LKFLOW_HAVE_OPENCV
#error "the OpenCV branch of lkflow.hpp was not taken"
#endif

static void fill(cv::Mat &m, int seed)
{
    for (int y = 0; y < m.rows; ++y)
        for (int x = 0; x < m.cols; ++x)
            m.at<double>(y, x) = 0.5 + 0.25 * std::sin(0.11 * (x + seed)) * std::cos(0.07 * (y - seed)) + 0.2 * std::sin(0.031 * x * y / 64.0);
}

int main(int argc, char **argv)
{
    if (argc != 2 || std::strcmp(argv[1], "run") != 0) { std::fprintf(stderr, "usage: %s run\n", argv[0]); return 2; }
    try {
        const int H = 64, W = 96;
        cv::Mat a(H, W, CV_64FC1), b(H, W, CV_64FC1);                       // continuous
        cv::Mat ap(H, W, CV_64FC1, (size_t)(W + 5) * 8), bp(H, W, CV_64FC1, (size_t)(W + 11) * 8);   // padded rows, two different steps
        fill(a, 1); fill(b, 2);
        for (int y = 0; y < H; ++y) { std::memcpy(ap.ptr<double>(y), a.ptr<double>(y), W * 8); std::memcpy(bp.ptr<double>(y), b.ptr<double>(y), W * 8); }
        cv::Mat u, v, up, vp;                                               // callers pass empty Mats (:822-823)
        lkflow::coarseToFineEstimation(a, b, u, v, lkflow::getMaxLayer(a));  // overwrites a, b (level-0 warp)
        lkflow::coarseToFineEstimation(ap, bp, up, vp, lkflow::getMaxLayer(ap));
        if (u.rows != H || u.cols != W || u.type() != CV_64FC1) { std::fprintf(stderr, "outputs not re-seated\n"); return 1; }
        double d = 0.0, dm = 0.0;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                d = std::fmax(d, std::fabs(u.at<double>(y, x) - up.at<double>(y, x)) + std::fabs(v.at<double>(y, x) - vp.at<double>(y, x)));
                dm = std::fmax(dm, std::fabs(a.at<double>(y, x) - ap.at<double>(y, x)) + std::fabs(b.at<double>(y, x) - bp.at<double>(y, x)));
            }
        cv::Mat u2, v2, u2p, v2p;
        lkflow::getLucasKanadeOpticalFlow(a, b, u2, v2);
        lkflow::getLucasKanadeOpticalFlow(ap, bp, u2p, v2p);
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) d = std::fmax(d, std::fabs(v2.at<double>(y, x) - v2p.at<double>(y, x)));
        bool threw = false;
        try { cv::Mat c(H, W + 2, CV_64FC1), uu, vv; lkflow::computeFlow(a, c, uu, vv, 1, 3); } catch (const std::invalid_argument &) { threw = true; }
        std::printf("max |d flow| = %g, max |d mutated inputs| = %g, size mismatch throws: %d\n", d, dm, (int)threw);
        if (d != 0.0 || dm != 0.0 || !threw) return 1;
        // Two back-to-back enqueue() calls WITHOUT a wait between them (the C ABI allows it: the second waits for the first),
        // unequal steps, LKFLOW_MUTATE_INPUTS, two different sizes (the engine's staging buffer must grow under the second
        // call), pageable and page-locked host buffers: flow and overwritten inputs must equal the synchronous calls'.
        for (int pin = 0; pin < 2; ++pin) {
            const int H2 = 96, W2 = 160;
            cv::Mat c1(H, W, CV_64FC1), c2(H, W, CV_64FC1, (size_t)(W + 7) * 8);                 // small pair, unequal steps
            cv::Mat e1(H2, W2, CV_64FC1), e2(H2, W2, CV_64FC1, (size_t)(W2 + 3) * 8);             // larger pair, unequal steps
            cv::Mat r1(H, W, CV_64FC1), r2(H, W, CV_64FC1), s1(H2, W2, CV_64FC1), s2(H2, W2, CV_64FC1);   // continuous twins
            fill(c1, 5); fill(r1, 5); fill(r2, 6); fill(e1, 7); fill(s1, 7); fill(s2, 8);
            for (int y = 0; y < H; ++y) std::memcpy(c2.ptr<double>(y), r2.ptr<double>(y), W * 8);
            for (int y = 0; y < H2; ++y) std::memcpy(e2.ptr<double>(y), s2.ptr<double>(y), W2 * 8);
            cv::Mat ur, vr, us, vs, uc, vc, ue, ve;
            lkflow::coarseToFineEstimation(r1, r2, ur, vr, 3);
            lkflow::coarseToFineEstimation(s1, s2, us, vs, 3);
            {
                lkflow::Engine eng(0);
                eng.pinHostBuffers(pin != 0);
                eng.enqueue(c1, c2, uc, vc, 3, 3, LKFLOW_MUTATE_INPUTS);
                eng.enqueue(e1, e2, ue, ve, 3, 3, LKFLOW_MUTATE_INPUTS);     // no wait() in between
                eng.wait();
                eng.pinHostBuffers(false);
            }
            double da = 0.0;
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x)
                    da = std::fmax(da, std::fabs(uc.at<double>(y, x) - ur.at<double>(y, x)) + std::fabs(vc.at<double>(y, x) - vr.at<double>(y, x)) +
                                           std::fabs(c1.at<double>(y, x) - r1.at<double>(y, x)) + std::fabs(c2.at<double>(y, x) - r2.at<double>(y, x)));
            for (int y = 0; y < H2; ++y)
                for (int x = 0; x < W2; ++x)
                    da = std::fmax(da, std::fabs(ue.at<double>(y, x) - us.at<double>(y, x)) + std::fabs(ve.at<double>(y, x) - vs.at<double>(y, x)) +
                                           std::fabs(e1.at<double>(y, x) - s1.at<double>(y, x)) + std::fabs(e2.at<double>(y, x) - s2.at<double>(y, x)));
            std::printf("back-to-back enqueue (pinned %d): max |d| = %g\n", pin, da);
            if (da != 0.0) return 1;
        }
        std::printf("ok\n");
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
