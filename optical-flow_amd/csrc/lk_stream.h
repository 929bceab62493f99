// lk_stream.h -- the frame-queue engine over a DEVICE LIST, behind the C ABI (host code; included by lkflow.hip).
//
// The reference's video loop (ParallelVideoPyr.cpp:769-873) takes frames one at a time: previous / current edge image ->
// coarseToFineEstimation (:832) -> |flow| threshold + centroid (:845-873).  north_star: "frames of a video stream shard
// naturally across the 8 GPUs of one node (frame-pair partition ...)".  This is that partition for a C / C++ caller that
// lives in ONE process (the multi-process form, one rank per GPU over RCCL, is optical-flow_amd/sharding.py):
//
//   lkflow_stream_create(devices, n, rows, cols, levels, win)   one worker = one lkflow_ctx + stream per list entry
//   lkflow_stream_push_u8(frames, F, ...)                       F 8-bit frames = F - 1 consecutive pairs, cut into contiguous
//                                                               blocks with a ONE-FRAME overlap (the rule of
//                                                               lkflow_amd.shard_pairs / SURVEY.md 8(e)); returns at once
//   lkflow_stream_centroids(out, n_pairs)                       waits; out[p] = {count, sum x, sum y} of pair p
//   lkflow_stream_flow(pair, U, V, step)                        optional: the flow of one pair (LKFLOW_STREAM_KEEP_FLOW)
//
// Per worker and push, everything on the worker's own stream: uint8 frames H2D (1 byte per pixel crosses PCIe), convertTo(CV_64F,
// 1/255) on the device (SerialPyr.cpp:237), lkflow_compute_batch_f64_dev on the block's consecutive pairs in chunks that fit
// the memory budget (a video batch: each frame's pyramid is built once), the fused centroid reduction, 3 doubles per pair D2H.
// Results do not depend on the device list: every pair's bits are those of the pair computed alone (the kernels' summation
// orders are anchored to image coordinates, not to the batch or strip geometry).
#pragma once

#include <thread>

struct lkflow_stream {
    struct Worker {
        int device = 0;
        lkflow_ctx *ctx = nullptr;
        unsigned char *d_u8 = nullptr; size_t u8_bytes = 0;      // the block's frames, 8-bit
        double *d_f64 = nullptr; size_t f64_bytes = 0;           // one chunk's frames as CV_64F
        double *d_uv = nullptr; size_t uv_bytes = 0;             // U planes then V planes (one chunk, or the whole block with KEEP_FLOW)
        double *d_cent = nullptr; size_t cent_bytes = 0;         // 3 doubles per pair of the block
        int first = 0, n_local = 0;                              // frames [first, first + n_local) of the last push
        int status = LKFLOW_OK;
        std::string error;
        std::thread th;
    };
    std::vector<Worker> w;
    int rows = 0, cols = 0, levels = 0, win = 0;
    unsigned flags = 0;
    int n_pairs = 0;                                            // of the last push
    std::vector<double> cent;                                   // [n_pairs][3]
    bool running = false;
    size_t budget_bytes = (size_t)16 << 30;                     // device memory a worker may hold per chunk (frames + flow + pyramid workspace)
    std::string last_error;
};

namespace {

// lkflow_amd.shard_pairs: contiguous blocks of the n_frames - 1 pairs, sizes differing by at most one, one-frame overlap
inline void stream_block(int n_frames, int world, int rank, int &first, int &n_local)
{
    const int n_pairs = n_frames > 1 ? n_frames - 1 : 0;
    const int base = n_pairs / world, rem = n_pairs % world;
    const int cnt = base + (rank < rem ? 1 : 0);
    first = rank * base + (rank < rem ? rank : rem);
    n_local = cnt > 0 ? cnt + 1 : 0;
}

inline int stream_grow(lkflow_ctx *ctx, void **p, size_t *have, size_t need)
{
    if (need <= *have) return LKFLOW_OK;
    if (*p) { HIP_TRY(ctx, hipFree(*p)); *p = nullptr; *have = 0; }
    hipError_t e = hipMalloc(p, need);
    if (e != hipSuccess) { hip_ok(ctx, e, "hipMalloc(stream block)"); *p = nullptr; return LKFLOW_E_NOMEM; }
    *have = need;
    return LKFLOW_OK;
}

// one worker, one push: runs on the worker's thread
int stream_work(lkflow_stream *s, lkflow_stream::Worker &w, const unsigned char *frames, size_t frame_stride, size_t step, double thresh)
{
    lkflow_ctx *ctx = w.ctx;
    int rc = set_device(ctx);
    if (rc) return rc;
    const int rows = s->rows, cols = s->cols;
    const size_t npx = (size_t)rows * cols, plane_b = npx * 8;
    const int pairs = w.n_local - 1;
    const bool keep = (s->flags & LKFLOW_STREAM_KEEP_FLOW) != 0;
    // pairs per lkflow_compute_batch_f64_dev call: (c + 1) CV_64F frames + 2c flow planes + the pyramid workspace (pyramids
    // (c + 1) / 3, warped planes 2c * 4/3, coarse flow 2c / 3 planes) must fit the budget
    int chunk = (int)std::min<size_t>((size_t)pairs, std::max<size_t>(1, s->budget_bytes / (plane_b * 7)));
    hipStream_t st = ctx->own_stream;
    if ((rc = stream_grow(ctx, (void **)&w.d_u8, &w.u8_bytes, (size_t)w.n_local * npx + 64))) return rc;
    if ((rc = stream_grow(ctx, (void **)&w.d_f64, &w.f64_bytes, (size_t)(chunk + 1) * plane_b))) return rc;
    if ((rc = stream_grow(ctx, (void **)&w.d_uv, &w.uv_bytes, (size_t)2 * (keep ? pairs : chunk) * plane_b))) return rc;
    if ((rc = stream_grow(ctx, (void **)&w.d_cent, &w.cent_bytes, (size_t)pairs * 3 * sizeof(double)))) return rc;
    // the block's frames: one strided copy when the frames are equally spaced (they are: one base + frame stride)
    for (int f = 0; f < w.n_local; ++f)
        HIP_TRY(ctx, hipMemcpy2DAsync(w.d_u8 + (size_t)f * npx, cols, frames + (size_t)(w.first + f) * frame_stride, step, cols, rows,
                                      hipMemcpyHostToDevice, st));
    for (int p0 = 0; p0 < pairs; p0 += chunk) {
        const int c = std::min(chunk, pairs - p0);
        if ((rc = lkflow_u8_to_f64_dev(ctx, w.d_u8 + (size_t)p0 * npx, w.d_f64, (size_t)(c + 1) * npx, st))) return rc;
        double *U = w.d_uv + (keep ? (size_t)p0 * npx : 0);
        double *V = U + (size_t)(keep ? pairs : chunk) * npx;
        // consecutive pairs of one frame buffer: next = prev + one frame (the library builds every frame's pyramid once)
        if ((rc = lkflow_compute_batch_f64_dev(ctx, w.d_f64, w.d_f64 + npx, plane_b, plane_b, c, rows, cols, (size_t)cols * 8, U, V,
                                               plane_b, (size_t)cols * 8, s->levels, s->win, 0, st)))
            return rc;
        if ((rc = lkflow_flow_centroid_f64_dev(ctx, U, V, plane_b, c, rows, cols, (size_t)cols * 8, thresh, w.d_cent + (size_t)p0 * 3, st)))
            return rc;
    }
    HIP_TRY(ctx, hipMemcpyAsync(s->cent.data() + (size_t)w.first * 3, w.d_cent, (size_t)pairs * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return LKFLOW_OK;
}

int stream_join(lkflow_stream *s)
{
    int rc = LKFLOW_OK;
    if (s->running) {
        for (auto &w : s->w)
            if (w.th.joinable()) w.th.join();
        s->running = false;
    }
    for (auto &w : s->w)
        if (w.status != LKFLOW_OK && rc == LKFLOW_OK) {
            rc = w.status;
            s->last_error = "device " + std::to_string(w.device) + ": " + w.error;
        }
    return rc;
}

}  // namespace

extern "C" {

int lkflow_stream_create(lkflow_stream **out, const int *devices, int n_devices, int rows, int cols, int pyr_levels, int win,
                         unsigned flags)
{
    if (!out) return LKFLOW_E_ARG;
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 1024 || rows < 1 || cols < 1 || (flags & ~LKFLOW_STREAM_KEEP_FLOW)) return LKFLOW_E_ARG;
    if ((unsigned long long)rows * (unsigned long long)cols >= (1ull << 31)) return LKFLOW_E_ARG;
    if (pyr_levels < 1 || pyr_levels > max_levels(cols, rows)) return LKFLOW_E_LEVELS;
    int rc = check_win(win);
    if (rc) return rc;
    lkflow_stream *s = new (std::nothrow) lkflow_stream();
    if (!s) return LKFLOW_E_NOMEM;
    s->rows = rows; s->cols = cols; s->levels = pyr_levels; s->win = win; s->flags = flags;
    s->w.resize((size_t)n_devices);
    for (int k = 0; k < n_devices; ++k) {
        s->w[(size_t)k].device = devices[k];
        if ((rc = lkflow_ctx_create(&s->w[(size_t)k].ctx, devices[k]))) { lkflow_stream_destroy(s); return rc; }
    }
    *out = s;
    return LKFLOW_OK;
}

void lkflow_stream_destroy(lkflow_stream *s)
{
    if (!s) return;
    (void)stream_join(s);
    for (auto &w : s->w) {
        if (!w.ctx) continue;
        (void)hipSetDevice(w.device);
        (void)hipStreamSynchronize(w.ctx->own_stream);
        if (w.d_u8) (void)hipFree(w.d_u8);
        if (w.d_f64) (void)hipFree(w.d_f64);
        if (w.d_uv) (void)hipFree(w.d_uv);
        if (w.d_cent) (void)hipFree(w.d_cent);
        lkflow_ctx_destroy(w.ctx);
    }
    delete s;
}

int lkflow_stream_set_budget(lkflow_stream *s, size_t bytes_per_device)
{
    if (!s || bytes_per_device < ((size_t)1 << 20)) return LKFLOW_E_ARG;
    const int rc = stream_join(s);
    s->budget_bytes = bytes_per_device;
    return rc;
}

int lkflow_stream_push_u8(lkflow_stream *s, const unsigned char *frames, int n_frames, size_t frame_stride_bytes, size_t step_bytes,
                          double thresh)
{
    if (!s || !frames || n_frames < 2 || n_frames > (1 << 24)) return LKFLOW_E_ARG;
    if (step_bytes < (size_t)s->cols || frame_stride_bytes < (size_t)(s->rows - 1) * step_bytes + (size_t)s->cols) return LKFLOW_E_ARG;
    int rc = stream_join(s);                                     // one push in flight: a second one first completes the first
    (void)rc;                                                    // (its status was the first push's to report; this push starts clean)
    s->n_pairs = n_frames - 1;
    s->cent.assign((size_t)s->n_pairs * 3, 0.0);
    s->last_error.clear();
    const int world = (int)s->w.size();
    for (int k = 0; k < world; ++k) {
        auto &w = s->w[(size_t)k];
        stream_block(n_frames, world, k, w.first, w.n_local);
        w.status = LKFLOW_OK; w.error.clear();
    }
    s->running = true;
    for (int k = 0; k < world; ++k) {
        auto *w = &s->w[(size_t)k];
        if (w->n_local < 2) continue;                            // more devices than pairs: this one idles
        try {
            w->th = std::thread([s, w, frames, frame_stride_bytes, step_bytes, thresh] {
                w->status = stream_work(s, *w, frames, frame_stride_bytes, step_bytes, thresh);
                if (w->status != LKFLOW_OK) {
                    w->error = w->ctx->last_error.empty() ? lkflow_strerror(w->status) : w->ctx->last_error;
                    (void)hipStreamSynchronize(w->ctx->own_stream);   // nothing of this worker stays in flight on the caller's frames
                }
            });
        } catch (const std::exception &e) {
            w->status = LKFLOW_E_NOMEM; w->error = e.what();
        }
    }
    return LKFLOW_OK;
}

int lkflow_stream_centroids(lkflow_stream *s, double *out, int n_pairs)
{
    if (!s || !out || n_pairs != s->n_pairs || n_pairs < 1) return LKFLOW_E_ARG;
    const int rc = stream_join(s);
    if (rc) return rc;
    std::memcpy(out, s->cent.data(), (size_t)n_pairs * 3 * sizeof(double));
    return LKFLOW_OK;
}

int lkflow_stream_flow(lkflow_stream *s, int pair, double *U, double *V, size_t uv_step_bytes)
{
    if (!s || !U || !V || pair < 0 || pair >= s->n_pairs) return LKFLOW_E_ARG;
    if (!(s->flags & LKFLOW_STREAM_KEEP_FLOW)) return LKFLOW_E_ARG;
    if (uv_step_bytes < (size_t)s->cols * 8 || uv_step_bytes % 8) return LKFLOW_E_ARG;
    int rc = stream_join(s);
    if (rc) return rc;
    for (auto &w : s->w) {
        const int pairs = w.n_local - 1;
        if (pairs < 1 || pair < w.first || pair >= w.first + pairs) continue;
        if ((rc = set_device(w.ctx))) return rc;
        const size_t npx = (size_t)s->rows * s->cols, row_b = (size_t)s->cols * 8;
        const double *dU = w.d_uv + (size_t)(pair - w.first) * npx, *dV = dU + (size_t)pairs * npx;
        HIP_TRY(w.ctx, hipMemcpy2D(U, uv_step_bytes, dU, row_b, row_b, (size_t)s->rows, hipMemcpyDeviceToHost));
        HIP_TRY(w.ctx, hipMemcpy2D(V, uv_step_bytes, dV, row_b, row_b, (size_t)s->rows, hipMemcpyDeviceToHost));
        return LKFLOW_OK;
    }
    return LKFLOW_E_ARG;
}

int lkflow_stream_block(int n_frames, int n_devices, int k, int *first_frame, int *n_frames_local)
{
    if (n_frames < 0 || n_devices < 1 || k < 0 || k >= n_devices || !first_frame || !n_frames_local) return LKFLOW_E_ARG;
    stream_block(n_frames, n_devices, k, *first_frame, *n_frames_local);
    return LKFLOW_OK;
}

const char *lkflow_stream_last_error(const lkflow_stream *s) { return s ? s->last_error.c_str() : ""; }

}  // extern "C"
