// dropin_check.cpp -- the reference's still-image harness (ParallelBasicPyr.cpp:717-767 and the
// golden generator SerialPyr.cpp:217-254) written against include/lkflow.hpp instead of OpenCV:
//   load table1/table2 crop -> convertTo(CV_64F, 1/255) -> getMaxLayer -> coarseToFineEstimation
//   -> getLucasKanadeOpticalFlow on the (now warped) inputs
// and dumps U, V, V2 as raw doubles for tests/test_cpp_dropin.py to compare with the goldens.
//   usage: dropin_check <table1.u8> <table2.u8> <out.f64>
#include <chrono>
#include <cstdio>
#include <vector>

#include "lkflow.hpp"

using lkflow::Mat;

static Mat load_u8_as_f64(const char *path, int rows, int cols)
{
    std::vector<unsigned char> buf((size_t)rows * cols);
    FILE *f = std::fopen(path, "rb");
    if (!f || std::fread(buf.data(), 1, buf.size(), f) != buf.size()) throw std::runtime_error(std::string("cannot read ") + path);
    std::fclose(f);
    Mat m(rows, cols);
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) m.at<double>(y, x) = (double)buf[(size_t)y * cols + x] * (1.0 / 255);  // SerialPyr.cpp:237
    return m;
}

int main(int argc, char **argv)
{
    if (argc != 4) { std::fprintf(stderr, "usage: %s table1.u8 table2.u8 out.f64\n", argv[0]); return 2; }
    try {
        Mat img1 = load_u8_as_f64(argv[1], 448, 640);
        Mat img2 = load_u8_as_f64(argv[2], 448, 640);
        const int maxLayer = lkflow::getMaxLayer(img1);
        std::printf("%d, %d, Max layer = %d\n", img1.rows, img1.cols, maxLayer);

        Mat u, v, u2, v2;                      // callers pass empty Mats (ParallelVideoPyr.cpp:822-823)
        const auto t0 = std::chrono::steady_clock::now();
        if (maxLayer >= 1) lkflow::coarseToFineEstimation(img1, img2, u, v, maxLayer);
        const auto t1 = std::chrono::steady_clock::now();
        std::printf("elapsed time = %f seconds.\n", std::chrono::duration<double>(t1 - t0).count());
        lkflow::getLucasKanadeOpticalFlow(img1, img2, u2, v2);      // SerialPyr.cpp:252

        // error behaviour: more levels than getMaxLayer must throw (reference: cv::Exception at :635)
        bool threw = false;
        try { Mat a = img1, b = img2, uu, vv; lkflow::coarseToFineEstimation(a, b, uu, vv, maxLayer + 1); }
        catch (const std::invalid_argument &) { threw = true; }
        if (!threw) { std::fprintf(stderr, "expected an exception for nLevels > getMaxLayer\n"); return 1; }

        FILE *f = std::fopen(argv[3], "wb");
        if (!f) throw std::runtime_error("cannot write output");
        for (const Mat *m : {&u, &v, &v2})
            for (int y = 0; y < m->rows; ++y) std::fwrite(m->ptr<double>(y), sizeof(double), m->cols, f);
        std::fclose(f);
        std::printf("Success!\n");
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
