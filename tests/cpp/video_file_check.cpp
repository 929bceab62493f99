// lkflow::VideoFile (include/lkingest.hpp) used the way the reference's loop uses cv::VideoCapture (ParallelVideoPyr.cpp:769-776):
// open, isOpened, read frame after frame until the stream ends.  usage: video_file_check in.mp4 out_luma.u8 out_first_bgr.u8
// Writes every frame's luma plane (tests/test_h264_ingest.py compares them with the Python decoder's fixture) and the first frame as BGR.
#include <cstdio>
#include <vector>

#include "lkingest.hpp"

static int run(int argc, char **argv);
int main(int argc, char **argv)
{
    try { return run(argc, argv); }
    catch (const std::exception &e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }   // a stream that does not decode: status, not abort
}

static int run(int argc, char **argv)
{
    if (argc != 4) { std::fprintf(stderr, "usage: %s in.mp4 out_luma.u8 out_first_bgr.u8\n", argv[0]); return 2; }
    lkflow::VideoFile missing("/nonexistent/file.mp4");
    if (missing.isOpened() || missing.status() != LKINGEST_E_IO) { std::fprintf(stderr, "a missing file must not open\n"); return 1; }
    lkflow::VideoFile cap(argv[1]);
    if (!cap.isOpened()) { std::fprintf(stderr, "cannot open %s (status %d)\n", argv[1], cap.status()); return 1; }
    std::printf("%d x %d, %d frames\n", cap.cols(), cap.rows(), cap.frameCount());
    std::vector<unsigned char> bgr;
    {   // the first frame as BGR from a second reader (cap >> frame)
        lkflow::VideoFile first(argv[1]);
        if (!first.read(bgr)) return 1;
        FILE *f = std::fopen(argv[3], "wb");
        if (!f || std::fwrite(bgr.data(), 1, bgr.size(), f) != bgr.size()) return 1;
        std::fclose(f);
    }
    std::vector<unsigned char> gray((size_t)cap.rows() * cap.cols());
    FILE *f = std::fopen(argv[2], "wb");
    if (!f) return 1;
    int n = 0;
    try {
        while (cap.readGray(gray.data(), (size_t)cap.cols())) {
            if (std::fwrite(gray.data(), 1, gray.size(), f) != gray.size()) return 1;
            ++n;
        }
    } catch (const std::exception &e) { std::fprintf(stderr, "frame %d: %s\n", n, e.what()); return 1; }
    std::fclose(f);
    if (cap.readGray(gray.data(), (size_t)cap.cols())) { std::fprintf(stderr, "a read after the end must fail\n"); return 1; }
    std::printf("frames read: %d\n", n);
    return n == cap.frameCount() ? 0 : 1;
}
