// stream_check.cpp -- the C-ABI frame-queue engine (lkflow_stream_*, lkflow::StreamEngine) used the way a C++ caller of the
// reference's video loop (ParallelVideoPyr.cpp:769-873) would: a stream of 8-bit frames in, one {count, sum x, sum y} row per
// frame pair out, pairs partitioned over a device list.  On a one-GPU box the list is {0, 0, 0}: three workers, three
// contexts, three streams on one device.  Checked against a SINGLE context walking the same stream through the plain C ABI
// (lkflow_u8_to_f64_dev + lkflow_compute_batch_f64_dev + lkflow_flow_centroid_f64_dev): rows and flows bit for bit.
//   usage: stream_check                 -> compile / link check only (exit 2)
//          stream_check run [n_devices] -> needs a GPU
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define LKFLOW_NO_OPENCV
#include "lkflow.hpp"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define CHECK_LK(x) do { int r_ = (x); if (r_ != LKFLOW_OK) { std::fprintf(stderr, "%s: %s\n", #x, lkflow_strerror(r_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 2 || std::strcmp(argv[1], "run") != 0) { std::fprintf(stderr, "usage: %s run [n_devices]\n", argv[0]); return 2; }
    const int ndev = argc > 2 ? std::atoi(argv[2]) : 3;
    const int H = 96, W = 160, F = 11, L = 3, win = 5;
    const double thresh = 0.75;
    // a textured blob that moves ~1 px per frame over a textured background, 8-bit
    std::vector<unsigned char> frames((size_t)F * H * W);
    for (int f = 0; f < F; ++f)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const double xs = x - 1.1 * f, ys = y + 0.6 * f;
                double v = 120.0 + 50.0 * std::sin(0.19 * xs) * std::cos(0.13 * ys) + 30.0 * std::sin(0.05 * (xs + 2 * ys));
                if ((x - 40 - 3 * f) * (x - 40 - 3 * f) + (y - 50) * (y - 50) < 400) v += 60.0 * std::cos(0.4 * (x - 3 * f)) * std::sin(0.3 * y);
                frames[((size_t)f * H + y) * W + x] = (unsigned char)std::lrint(std::fmin(std::fmax(v, 0.0), 255.0));
            }
    try {
        // ---- reference run: one context, the whole stream in one call
        const size_t npx = (size_t)H * W;
        std::vector<double> U0((F - 1) * npx), V0((F - 1) * npx), C0((size_t)(F - 1) * 3);
        {
            lkflow_ctx *ctx = nullptr;
            CHECK_LK(lkflow_ctx_create(&ctx, 0));
            unsigned char *d8; double *d64, *dU, *dV, *dC;
            CHECK_HIP(hipMalloc((void **)&d8, F * npx)); CHECK_HIP(hipMalloc((void **)&d64, F * npx * 8));
            CHECK_HIP(hipMalloc((void **)&dU, (F - 1) * npx * 8)); CHECK_HIP(hipMalloc((void **)&dV, (F - 1) * npx * 8));
            CHECK_HIP(hipMalloc((void **)&dC, (F - 1) * 3 * 8));
            CHECK_HIP(hipMemcpy(d8, frames.data(), F * npx, hipMemcpyHostToDevice));
            CHECK_LK(lkflow_u8_to_f64_dev(ctx, d8, d64, F * npx, nullptr));
            CHECK_LK(lkflow_compute_batch_f64_dev(ctx, d64, d64 + npx, npx * 8, npx * 8, F - 1, H, W, (size_t)W * 8, dU, dV, npx * 8,
                                                  (size_t)W * 8, L, win, 0, nullptr));
            CHECK_LK(lkflow_flow_centroid_f64_dev(ctx, dU, dV, npx * 8, F - 1, H, W, (size_t)W * 8, thresh, dC, nullptr));
            CHECK_HIP(hipDeviceSynchronize());
            CHECK_HIP(hipMemcpy(U0.data(), dU, (F - 1) * npx * 8, hipMemcpyDeviceToHost));
            CHECK_HIP(hipMemcpy(V0.data(), dV, (F - 1) * npx * 8, hipMemcpyDeviceToHost));
            CHECK_HIP(hipMemcpy(C0.data(), dC, (F - 1) * 3 * 8, hipMemcpyDeviceToHost));
            for (void *q : {(void *)d8, (void *)d64, (void *)dU, (void *)dV, (void *)dC}) CHECK_HIP(hipFree(q));
            lkflow_ctx_destroy(ctx);
        }
        double moved = 0.0;
        for (int p = 0; p < F - 1; ++p) moved += C0[(size_t)p * 3];
        if (!(moved > 100.0)) { std::fprintf(stderr, "the test stream does not move (%g pixels over the threshold)\n", moved); return 1; }

        // ---- the engine over a device list, several list lengths (more workers than pairs included), tiny memory budget
        for (int n : {1, ndev, 4, 16}) {
            std::vector<int> devs((size_t)n, 0);
            lkflow::StreamEngine eng(devs, H, W, L, win, /*keep_flow=*/true);
            if (n == 4) CHECK_LK(lkflow_stream_set_budget(eng.handle(), (size_t)1 << 20));   // one pair per call inside a block
            eng.push(frames.data(), F, thresh);
            const std::vector<double> C = eng.centroids();
            if (std::memcmp(C.data(), C0.data(), C0.size() * 8) != 0) { std::fprintf(stderr, "centroid rows differ (%d workers)\n", n); return 1; }
            lkflow::Mat U, V;
            for (int p : {0, 3, F - 2}) {
                eng.flow(p, U, V);
                for (int y = 0; y < H; ++y)
                    if (std::memcmp(U.ptr<double>(y), &U0[(size_t)p * npx + (size_t)y * W], (size_t)W * 8) != 0 ||
                        std::memcmp(V.ptr<double>(y), &V0[(size_t)p * npx + (size_t)y * W], (size_t)W * 8) != 0) {
                        std::fprintf(stderr, "flow of pair %d differs (%d workers)\n", p, n); return 1;
                    }
            }
            // a second push on the same engine (buffers re-used), shorter stream
            eng.push(frames.data() + npx, F - 4, thresh);
            const std::vector<double> C2 = eng.centroids();
            if (std::memcmp(C2.data(), C0.data() + 3, C2.size() * 8) != 0) { std::fprintf(stderr, "second push differs (%d workers)\n", n); return 1; }
            const lkflow::Gesture g = lkflow::gestureCentroid(C[0], C[1], C[2], W, W, 50.0);
            std::printf("%2d worker(s): %d pairs bit-identical; pair 0: counts %.0f -> circle at (%.1f, %.1f) detected %d\n", n, F - 1, C[0], g.x, g.y, (int)g.detected);
        }
        // ---- partition rule and argument errors
        int first = -1, nl = -1, covered = 0;
        for (int k = 0; k < 3; ++k) { CHECK_LK(lkflow_stream_block(F, 3, k, &first, &nl)); if (first != covered) return 1; covered += nl - 1; }
        if (covered != F - 1) { std::fprintf(stderr, "blocks do not cover the pairs\n"); return 1; }
        bool threw = false;
        try { lkflow::StreamEngine bad({0}, H, W, 9, win); } catch (const std::invalid_argument &) { threw = true; }   // too many levels
        if (!threw) { std::fprintf(stderr, "level check missing\n"); return 1; }
        threw = false;
        try { lkflow::StreamEngine bad({99}, H, W, L, win); } catch (const std::runtime_error &) { threw = true; }      // no such device
        if (!threw) { std::fprintf(stderr, "device check missing\n"); return 1; }
        std::printf("ok\n");
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
