// CPU-only check of the header-only pieces of include/lkflow.hpp that need no GPU: the host tail of the flow
// consumer (ParallelVideoPyr.cpp:875-888) and the Mat stand-in.  Exit code 0 = all checks passed.
#define LKFLOW_NO_OPENCV
#include "lkflow.hpp"

#include <cstdio>

int main()
{
    using namespace lkflow;
    int bad = 0;
    // counts > 500 decides; the scale is the INTEGER ratio frame.cols / current_frame.cols (:880)
    Gesture g = gestureCentroid(501.0, 501.0 * 100.0, 501.0 * 50.0, 1920, 1080);
    bad += !(g.detected && g.x == 100.0 && g.y == 50.0);              // 1920 / 1080 == 1
    g = gestureCentroid(500.0, 500.0 * 10.0, 500.0 * 20.0, 1920, 1080);
    bad += !(!g.detected && g.x == 10.0 && g.y == 20.0);              // exactly 500: no circle
    g = gestureCentroid(1000.0, 1000.0 * 7.0, 1000.0 * 3.0, 2160, 1080);
    bad += !(g.detected && g.x == 14.0 && g.y == 6.0);                // 2160 / 1080 == 2
    g = gestureCentroid(0.0, 0.0, 0.0, 1920, 1080);
    bad += !(!g.detected && g.x == 0.0 && g.y == 0.0);                // the reference divides 0/0 here and draws nothing
    Mat m = Mat::zeros(3, 5);
    m.at<double>(2, 4) = 1.5;
    Mat c = m;
    bad += !(c.rows == 3 && c.cols == 5 && c.step == 40 && c.at<double>(2, 4) == 1.5 && c.data != m.data);
    std::printf("%s\n", bad ? "FAILED" : "ok");
    return bad;
}
