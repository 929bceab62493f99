// lkflow.hip -- host side of liblkflow.so: context, workspace, launch sequencing, C ABI.
//
// The coarse-to-fine driver below is the device counterpart of coarseToFineEstimation
// (ParallelVideoPyr.cpp:578-725 / SerialPyr.cpp:157-194):
//     pyramids (a2) -> for l = L-1 .. 0 { [warp with up(flow_{l+1})] (a9-a12) ; LK + up (a3-a8) }
// Everything is enqueued on ONE stream with no host synchronisation, so a caller may capture a
// call into a HIP graph after lkflow_ctx_reserve().
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/lkflow.h"
#include "lk_kernels.h"
#include "lk_march3.h"
#include "lk_split.h"
#include "lk_w3x2.h"
#include "lk_w3x1f.h"
// Kernels that were measured SLOWER than the shipped path and are kept as records: the batch-synchronous
// kernel (lk_batch.h, LKFLOW_OPT_LK_BATCH) and the two-column fused 3 x 3 kernel (lk_w3x2f.h).  Only a -DLKFLOW_EXPERIMENTAL
// build (tools/build_variant.sh) carries them; the default library answers LKFLOW_E_ARG to their options.
#ifdef LKFLOW_EXPERIMENTAL
#include "lk_batch.h"
#include "lk_w3x2f.h"
#endif

// the digest of the kernel sources this library was built from (__graft_entry__.csrc_digest(), passed by build_hip()): bench.py
// and tools/pmc_summary.py attach committed profile numbers only to a LOADED library that carries the digest they were measured on
#ifndef LKFLOW_CSRC_SHA
#define LKFLOW_CSRC_SHA "unstamped"
#endif
#define LKFLOW_VERSION_STR "lkflow-mi355x 0.4 (gfx950) src:" LKFLOW_CSRC_SHA

namespace {

// A/B knobs of the tuning tools (tools/ab.sh builds a second library with -DLKFLOW_DEBUG_KNOBS and points the Python
// binding at it).  The shipped library reads NO environment variable: every knob folds to its default at compile time.
#ifdef LKFLOW_DEBUG_KNOBS
inline int knob(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
#else
constexpr int knob(const char *, int dflt) { return dflt; }
#endif
// "this launcher does not take the shape" (the caller falls back to the next kernel); distinct from every lkflow_status
constexpr int LK_NOT_TAKEN = 1;

using lk::CPlanes;
using lk::Planes;

inline long long align_up(long long v, long long a) { return (v + a - 1) / a * a; }

struct Level {
    int H, W;
    long long pitch;  // elements
    long long plane;  // elements (pitch * H, padded)
};

}  // namespace

#ifndef LKFLOW_SPLIT_FUSED_DEFAULT
#define LKFLOW_SPLIT_FUSED_DEFAULT 2                 /* LKFLOW_OPT_FUSE_WARP: 0 never, 1 wherever the kernel takes the shape, 2 where it is faster */
#endif
struct lkflow_ctx {
    int device = 0;
    int n_cu = 256;
    hipStream_t own_stream = nullptr;
    // fork / join for the clipped-window frame kernel, which runs beside the main LK kernel (created on first use)
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::string last_error;
    // options
    int impl = 0;
    int fuse_warp = LKFLOW_SPLIT_FUSED_DEFAULT;   // LKFLOW_OPT_FUSE_WARP
    int lk_batch = 0;                             // LKFLOW_OPT_LK_BATCH
    int tile_rows = 0;
    int profile = 0;
    struct Rec { hipEvent_t a, b; int slot; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> ev_pool;
    // workspace arena
    char *arena = nullptr;
    size_t arena_bytes = 0;
    // geometry the arena was carved for
    int g_rows = 0, g_cols = 0, g_levels = 0, g_pairs = 0;
    std::vector<Level> lv;
    // arena offsets (in elements) per level; planes of frame 1 then frame 2: [2*pairs] planes
    std::vector<long long> off_pyr;   // level >= 1
    std::vector<long long> off_warp;  // level <= L-2
    std::vector<long long> off_flow;  // level >= 1: u planes [pairs] then v planes [pairs]
    // resident blocks per CU of each kernel instantiation ON THIS DEVICE (occupancy query, first use)
    std::unordered_map<const void *, int> occ;
    // launch plans of the wave-specialised kernel by (rows, halo, column strips, slots, lone slots, cap): a video loop asks for the
    // same few again and again, and the planner's search grows with rows x column strips / slots
    struct Plan { int H, R; long long nblk, slots, lone, cap; int thA; long long csA; int thB; };
    std::vector<Plan> plans;
    // host-API staging (prev, next, U, V contiguous planes [+ two uint8 planes])
    double *stage = nullptr;
    size_t stage_bytes = 0;
    bool host_inflight = false;        // an enqueued host-buffer pair has not been waited for yet
    // opt-in (LKFLOW_OPT_PIN_HOST): caller buffers page-locked in place, so that the host <-> device copies are
    // true asynchronous DMA; kept until the context is destroyed or the option is switched off
    int pin_host = 0;
    struct Pin { char *p; size_t n; };
    std::vector<Pin> pins;
};

namespace {

bool hip_ok(lkflow_ctx *ctx, hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    if (ctx) ctx->last_error = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}
// Release one LKFLOW_OPT_PIN_HOST registration.  A range that another context (or the caller) registered or released
// in the meantime makes the runtime report an error; it must not stay behind as the thread's sticky "last error",
// where the next unrelated HIP call of the caller would find it.
inline void unpin_host_range(void *p)
{
    if (hipHostUnregister(p) != hipSuccess) (void)hipGetLastError();
}
#define HIP_TRY(ctx, call)                                   \
    do {                                                     \
        if (!hip_ok((ctx), (call), #call)) return LKFLOW_E_HIP; \
    } while (0)

// ---- optional per-launch HIP-event bracketing (LKFLOW_OPT_PROFILE) ------------------------------
constexpr size_t MAX_PROFILE_RECS = 1u << 16;   // launches bracketed between two profile reads; further ones go untimed

hipEvent_t ev_get(lkflow_ctx *ctx)                // nullptr: no event could be created (the launch then goes untimed)
{
    if (!ctx->ev_pool.empty()) { hipEvent_t e = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
struct ProfScope {
    lkflow_ctx *ctx; hipStream_t s; bool on; hipEvent_t a = nullptr, b = nullptr; int slot;
    ProfScope(lkflow_ctx *c, int stage, int level, hipStream_t st) : ctx(c), s(st), slot(stage * LKFLOW_MAX_LEVELS + level)
    {
        on = stage >= 0 && (c->profile == -1 || c->profile == 1 + slot) && c->recs.size() < MAX_PROFILE_RECS;   // stage < 0: the caller brackets
        if (on) {
            a = ev_get(c); b = ev_get(c);
            if (!a || !b) {                                  // event creation failed: give back what exists, do not time
                if (a) c->ev_pool.push_back(a);
                if (b) c->ev_pool.push_back(b);
                a = b = nullptr; on = false;
            } else (void)hipEventRecord(a, s);
        }
    }
    ~ProfScope()
    {
        if (on) { (void)hipEventRecord(b, s); ctx->recs.push_back({a, b, slot}); }
    }
};

int max_levels(int cols, int rows)
{
    if (cols <= 0 || rows <= 0) return 0;
    int p = 1;
    while (cols % (1 << p) == 0) ++p;
    int res = p;
    p = 1;
    while (rows % (1 << p) == 0) ++p;
    return res < p ? res : p;
}

int check_common(int rows, int cols, size_t step_bytes, size_t uv_step_bytes, int n_pairs)
{
    if (rows < 1 || cols < 1 || n_pairs < 1 || n_pairs > 32767) return LKFLOW_E_ARG;
    if (step_bytes < (size_t)cols * 8 || step_bytes % 8) return LKFLOW_E_ARG;
    if (uv_step_bytes < (size_t)cols * 8 || uv_step_bytes % 8) return LKFLOW_E_ARG;
    // kernels index one plane with 32-bit element offsets
    if ((unsigned long long)rows * (step_bytes / 8) >= (1ull << 31)) return LKFLOW_E_ARG;
    if ((unsigned long long)rows * (uv_step_bytes / 8) >= (1ull << 31)) return LKFLOW_E_ARG;
    return LKFLOW_OK;
}

int check_win(int win)
{
    if (win < 3 || (win & 1) == 0 || win > LKFLOW_MAX_WIN) return LKFLOW_E_WIN;
    return LKFLOW_OK;
}

// Carve (or re-carve) the arena for a (rows, cols, levels, pairs) problem.
int ensure_workspace(lkflow_ctx *ctx, int rows, int cols, int levels, int pairs)
{
    if (ctx->g_rows == rows && ctx->g_cols == cols && ctx->g_levels == levels &&
        ctx->g_pairs >= pairs && ctx->arena)
        return LKFLOW_OK;
    std::vector<Level> lv(levels);
    lv[0] = {rows, cols, align_up(cols, 16), 0};
    for (int l = 1; l < levels; ++l)
        lv[l] = {(lv[l - 1].H + 1) / 2, (lv[l - 1].W + 1) / 2, 0, 0};
    for (int l = 0; l < levels; ++l) {
        lv[l].pitch = align_up(lv[l].W, 16);
        lv[l].plane = align_up(lv[l].pitch * lv[l].H, 32);
    }
    std::vector<long long> off_pyr(levels, -1), off_warp(levels, -1), off_flow(levels, -1);
    long long cur = 0;
    for (int l = 1; l < levels; ++l) { off_pyr[l] = cur; cur += 2LL * pairs * lv[l].plane; }
    for (int l = 0; l + 1 < levels; ++l) { off_warp[l] = cur; cur += 2LL * pairs * lv[l].plane; }
    for (int l = 1; l < levels; ++l) { off_flow[l] = cur; cur += 2LL * pairs * lv[l].plane; }
    size_t need = (size_t)cur * sizeof(double) + 256;
    if (need > ctx->arena_bytes) {
        HIP_TRY(ctx, hipDeviceSynchronize());
        if (ctx->arena) { HIP_TRY(ctx, hipFree(ctx->arena)); ctx->arena = nullptr; ctx->arena_bytes = 0; }
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, need);
        if (e != hipSuccess) {
            hip_ok(ctx, e, "hipMalloc(workspace)");
            return LKFLOW_E_NOMEM;
        }
        ctx->arena = (char *)p;
        ctx->arena_bytes = need;
    }
    ctx->lv = lv;
    ctx->off_pyr = off_pyr;
    ctx->off_warp = off_warp;
    ctx->off_flow = off_flow;
    ctx->g_rows = rows; ctx->g_cols = cols; ctx->g_levels = levels; ctx->g_pairs = pairs;
    return LKFLOW_OK;
}

inline double *arena_at(lkflow_ctx *ctx, long long off_elems)
{
    return reinterpret_cast<double *>(ctx->arena) + off_elems;
}

inline dim3 grid_px(int W, int H, int planes) { return dim3((W + 63) / 64, (H + 3) / 4, planes); }

// ---- LK launch ---------------------------------------------------------------------------------

// Resident blocks per CU of one kernel instantiation on the context's device (occupancy query, cached in the
// context: contexts are per device and per caller thread, so there is no shared mutable state).
template <typename Kern>
int blocks_per_cu(lkflow_ctx *ctx, Kern kern, int threads)
{
    const void *key = reinterpret_cast<const void *>(kern);
    auto it = ctx->occ.find(key);
    if (it != ctx->occ.end()) return it->second;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, threads, 0) != hipSuccess || n < 1) n = 1;
    ctx->occ.emplace(key, n);
    return n;
}

// Rows per block: every block marches th + 2R rows and blocks run in rounds of `slots` resident
// blocks -> minimise rounds * (th + 2R + fixed cost) over the number of row strips.
// lone_cus > 0 (the wave-specialised kernel at two workgroups per CU): a last, partial round that leaves at most one
// workgroup on every CU costs 0.62 of a full one -- a workgroup that has its CU to itself steps 1.6 times as fast as one
// of a pair (one workgroup per CU = +25 % time for the same work).  Measured on 32 x 1080p, 15 x 15
// (tools/sweep_lk_th2.py, profiles/r02_i_lk_strip_height_repeat.txt): 4.5 rounds of 270-row strips 0.645 ms against
// 0.671 ms for the 9 whole rounds of 136-row strips that the whole-rounds rule picks.
inline int pick_tile_rows(int H, int R, long long blocks_per_strip, long long slots, long long lone_cus = 0, long long *cost_out = nullptr)
{
    long long best = -1;
    int th = H;
    for (int gy = 1; gy <= H; ++gy) {
        const int t = (H + gy - 1) / gy;
        if (t < 6 && gy > 1) break;       // tiny strips only ever win when the grid is far below one round
        if ((H + t - 1) / t != gy) continue;
        const long long nblk = blocks_per_strip * gy;
        const long long full = nblk / slots, rem = nblk % slots;
        const long long rounds100 = full * 100 + (rem == 0 ? 0 : (rem <= lone_cus ? 62 : 100));
        const long long cost = rounds100 * (t + 2 * R + 6);
        if (best < 0 || cost < best) { best = cost; th = t; }
    }
    if (cost_out) *cost_out = best;
    return th;
}

// Two strip heights in one launch (the wave-specialised kernel's 1-D grid; lk_kernels.h split_tile).  Workgroups are dispatched
// in grid order into the slots earlier ones free, so a launch runs in rounds, and with ONE height the last round is either
// partly empty or paid for with short strips everywhere (each strip marches 2R + ~6 rows more than it stores: 32 x 1080p at
// 15 x 15 fused = 576 column strips on 256 slots -> 9 rounds of 270-row strips = 2646 marched rows per slot against 2430 stored).
// Plan: the first csA column strips in tall strips that fill k whole rounds exactly, the remaining nblk - csA in short strips
// that fill the last round(s) (2 x 1104 + 294 = 2502).  Same cost model as pick_tile_rows; one height when that is cheaper,
// when the user fixed the strip height, or when a tall strip would exceed `cap` rows.  Results do not depend on the partition
// (tests/test_gpu_round2.py test_every_window_multi_strip_multi_block).
struct StripPlan { int thA; long long csA; int thB; };
inline StripPlan pick_strip_plan(int H, int R, long long nblk, long long slots, long long lone_cus, long long cap)
{
    static const int two = knob("LKFLOW_TWO_HEIGHTS", 1);
    long long best = 0;
    int th = pick_tile_rows(H, R, nblk, slots, lone_cus, &best);
    th = (th + 1) & ~1;
    StripPlan p{th, nblk, th};
    if (!two || (slots & 7)) return p;
    for (int gyA = 1; gyA <= 8; ++gyA) {
        const int tA = ((H + gyA - 1) / gyA + 1) & ~1;
        if ((H + tA - 1) / tA != gyA || tA > cap) continue;
        for (long long k = 1; k * slots / gyA < nblk; ++k) {
            if ((k * slots) % gyA) continue;
            const long long csA = k * slots / gyA;
            long long cB = 0;
            const int tB = (pick_tile_rows(H, R, nblk - csA, slots, lone_cus, &cB) + 1) & ~1;
            if (tB >= tA) continue;
            const long long cost = k * 100 * (tA + 2 * R + 6) + cB;
            if (cost < best) { best = cost; p = StripPlan{tA, csA, tB}; }
        }
    }
    return p;
}

inline StripPlan cached_strip_plan(lkflow_ctx *ctx, int H, int R, long long nblk, long long slots, long long lone_cus, long long cap)
{
    for (const auto &q : ctx->plans)
        if (q.H == H && q.R == R && q.nblk == nblk && q.slots == slots && q.lone == lone_cus && q.cap == cap) return StripPlan{q.thA, q.csA, q.thB};
    const StripPlan p = pick_strip_plan(H, R, nblk, slots, lone_cus, cap);
    if (ctx->plans.size() >= 64) ctx->plans.clear();
    ctx->plans.push_back(lkflow_ctx::Plan{H, R, nblk, slots, lone_cus, cap, p.thA, p.csA, p.thB});
    return p;
}

// fills the tile fields of the wave-specialised kernel's arguments from a plan; returns the number of marching workgroups
// (0: too many for a 32-bit grid)
inline unsigned apply_strip_plan(lk::LKArgs &a, const StripPlan &p, int gx, long long nblk)
{
    a.gx = gx;
    a.th = p.thA; a.gy = (a.H + p.thA - 1) / p.thA;
    a.thB = p.thB; a.gyB = (a.H + p.thB - 1) / p.thB;
    const long long tA = p.csA * a.gy, tB = (nblk - p.csA) * a.gyB;
    if (tA + tB > 0x7fffffffLL) return 0;
    a.csA = (unsigned)p.csA; a.csB = (unsigned)(nblk - p.csA);
    a.tilesA = (unsigned)tA;
    return (unsigned)(tA + tB);
}

template <int WIN, int TX>
int launch_lk3_win(lkflow_ctx *ctx, lk::LKArgs a, bool has_up, int pairs, hipStream_t s)
{
    constexpr int R = WIN / 2;
    constexpr int xo = TX - 2 * R;
    const int gx = (a.W + xo - 1) / xo;
    int th = ctx->tile_rows;
    if (th <= 0) {
        const int bpc = has_up ? blocks_per_cu(ctx, lk::lk_march3_kernel<WIN, TX, true>, TX)
                               : blocks_per_cu(ctx, lk::lk_march3_kernel<WIN, TX, false>, TX);
        th = pick_tile_rows(a.H, R, (long long)gx * pairs, (long long)ctx->n_cu * bpc);
    }
    if (th > a.H) th = a.H;
    {   // the kernel addresses a strip with 32-bit byte offsets from its first row: cap the strip height
        long long pmax = a.i1.pitch;
        if (a.u.pitch > pmax) pmax = a.u.pitch;
        if (a.v.pitch > pmax) pmax = a.v.pitch;
        const long long cap = 0x7fffffffLL / (pmax * 8) - 2 * R - 8;
        if (cap < 1) return LKFLOW_E_ARG;
        if (th > cap) th = (int)cap;
    }
    a.th = th;
    dim3 grid(gx, (a.H + th - 1) / th, pairs);
    if (has_up)
        hipLaunchKernelGGL((lk::lk_march3_kernel<WIN, TX, true>), grid, dim3(TX), 0, s, a);
    else
        hipLaunchKernelGGL((lk::lk_march3_kernel<WIN, TX, false>), grid, dim3(TX), 0, s, a);
    return LKFLOW_OK;
}

// The frame of width R just inside the image (clipped windows, SerialPyr.cpp:94-95, and the outermost ring) is written
// by lk_border_kernel and by nobody else, so it does not have to wait for the main LK kernel: it is forked onto the
// context's side stream (both kernels depend on the warp before them only) and joined after the main launch.  The
// fork / join is a pair of events, so it is captured into a HIP graph like any other dependency.
// Only the -DLKFLOW_DEBUG_KNOBS build can take this path (LKFLOW_BORDER_MERGED=0 LKFLOW_BORDER_SIDE=1, A/B): the shipped
// library writes the frame in trailing workgroups of the main launch.
struct BorderFork {
    lkflow_ctx *ctx; hipStream_t s, bs;
    BorderFork(lkflow_ctx *c, hipStream_t st) : ctx(c), s(st), bs(st)
    {
        static const int side = knob("LKFLOW_BORDER_SIDE", 0);
        if (!side) return;
        if (!ctx->side_stream) {
            if (hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError();
                if (ctx->ev_fork) { (void)hipEventDestroy(ctx->ev_fork); ctx->ev_fork = nullptr; }
                if (ctx->side_stream) { (void)hipStreamDestroy(ctx->side_stream); ctx->side_stream = nullptr; }
                return;                                      // no side stream: the frame kernel runs in stream order
            }
        }
        if (hipEventRecord(ctx->ev_fork, s) != hipSuccess || hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0) != hipSuccess) {
            (void)hipGetLastError();
            return;
        }
        bs = ctx->side_stream;
    }
    // after the frame kernel has been launched on bs and the main kernel on s
    bool join()
    {
        if (bs == s) return true;
        return hipEventRecord(ctx->ev_join, bs) == hipSuccess && hipStreamWaitEvent(s, ctx->ev_join, 0) == hipSuccess;
    }
};

// Column blocks of the wave-specialised kernel and the column its block 0 starts its outputs at (LKArgs::xorg): the marching
// blocks cover the columns R .. W - R - 1 only (the frame of width R is the frame workgroups'), from the even column R & ~1 on.
template <typename C>
int split_column_blocks(lk::LKArgs &a)
{
    static const int shifted = knob("LKFLOW_SPLIT_XORG", 1);   // (A/B knob of a debug-knob build: 0 = blocks from column 0)
    a.xorg = (shifted && a.W > 2 * C::R) ? (C::R & ~1) : 0;
    const int need = a.xorg ? a.W - C::R - a.xorg : a.W;       // output columns xorg .. W - R - 1
    return std::max(1, (need + C::XO - 1) / C::XO);
}

// Wave-specialised kernel (lk_split.h): K*64-column blocks, K PE waves + 5 window-sum waves
template <int WIN, int K>
int launch_split_win(lkflow_ctx *ctx, lk::LKArgs a, bool has_up, int pairs, hipStream_t s)
{
    using C = lk::SplitCfg<WIN, K>;
    if ((((a.W | a.H) & 1) && has_up) || a.W < 2 || a.H <= C::R + C::PF + 1) return LK_NOT_TAKEN;   // tiny levels (odd ones under a coarser level do not exist): one-role kernel
    const int gx = split_column_blocks<C>(a);
    const long long pmax = std::max(a.i1.pitch, std::max(a.u.pitch, a.v.pitch));
    const long long cap = (0x7fffffffLL / (pmax * 8) - 2 * C::R - 8) & ~1LL;   // (32-bit byte offsets from a strip's first row)
    if (cap < 2) return LKFLOW_E_ARG;
    const long long nblk = (long long)gx * pairs;
    StripPlan plan;
    if (ctx->tile_rows > 0) {
        const int th = (int)std::min<long long>((ctx->tile_rows + 1) & ~1, cap);   // even strips: row parity is static in the unrolled pipeline
        plan = StripPlan{th, nblk, th};
    } else {
        const int bpc = has_up ? blocks_per_cu(ctx, lk::lk_split_kernel<WIN, K, true>, C::NT)
                               : blocks_per_cu(ctx, lk::lk_split_kernel<WIN, K, false>, C::NT);
        plan = cached_strip_plan(ctx, a.H, C::R + 1, nblk, (long long)ctx->n_cu * bpc, bpc == 2 ? ctx->n_cu : 0, cap);
        if (plan.thA > cap) plan = StripPlan{(int)cap, nblk, (int)cap};
    }
    const unsigned nmarch = apply_strip_plan(a, plan, gx, nblk);
    if (!nmarch) return LKFLOW_E_ARG;
    dim3 grid(nmarch);                               // 1-D: the kernel orders the tiles per XCD
    // the frame of clipped-window pixels (SerialPyr.cpp:94-95) and the outermost ring: frame pixels per pair
    const long long nr = std::min(2 * C::R, a.H), nc = std::min(2 * C::R, a.W);
    const long long nfp = nr * a.W + (a.H > 2 * C::R ? (long long)(a.H - 2 * C::R) * nc : 0);
    if (nfp + 2 * C::NT >= 0x7fffffffLL) return LKFLOW_E_ARG;   // (the frame code indexes the pixels of one plane's frame with 32 bits)
    static const int merged = knob("LKFLOW_BORDER_MERGED", 1);
    const long long nbb = (nfp + C::NT - 1) / C::NT;
    if (merged && (long long)grid.x + nbb * pairs <= 0x7fffffffLL) {
        // as trailing workgroups of the main launch
        a.nmain = grid.x; a.nbb = (unsigned)nbb;
        grid.x += (unsigned)(nbb * pairs);
        if (has_up) hipLaunchKernelGGL((lk::lk_split_kernel<WIN, K, true>), grid, dim3(C::NT), 0, s, a);
        else hipLaunchKernelGGL((lk::lk_split_kernel<WIN, K, false>), grid, dim3(C::NT), 0, s, a);
        return LKFLOW_OK;
    }
    a.nmain = 0; a.nbb = 0;
    BorderFork bf(ctx, s);
    {   // as a launch of its own, in front of the main kernel (or beside it, on the side stream)
        dim3 bg((unsigned)((nfp + 255) / 256), pairs);
        if (has_up) hipLaunchKernelGGL((lk::lk_border_kernel<true>), bg, dim3(256), 0, bf.bs, a, WIN);
        else hipLaunchKernelGGL((lk::lk_border_kernel<false>), bg, dim3(256), 0, bf.bs, a, WIN);
    }
    if (has_up) hipLaunchKernelGGL((lk::lk_split_kernel<WIN, K, true>), grid, dim3(C::NT), 0, s, a);
    else hipLaunchKernelGGL((lk::lk_split_kernel<WIN, K, false>), grid, dim3(C::NT), 0, s, a);
    if (!bf.join()) { hip_ok(ctx, hipGetLastError(), "frame kernel join"); return LKFLOW_E_HIP; }
    return LKFLOW_OK;
}

// Warp fused into the wave-specialised kernel (lk_split.h, W role): a.i1 / a.i2 are the UN-warped level images,
// the warped planes never reach memory.  LK_NOT_TAKEN: shape not taken (the caller runs warp + LK).
template <int WIN, int K>
int launch_splitw_win(lkflow_ctx *ctx, lk::LKArgs a, int pairs, hipStream_t s, int level)
{
    using C = lk::SplitCfg<WIN, K, true>;
    if ((a.W & 1) || (a.H & 1) || a.W < 2 || a.H <= C::R + C::PF + 1) return LK_NOT_TAKEN;
    // the W role addresses a whole image plane with 32-bit byte offsets and one pitch for both images
    // (forms row * pitch with a 24-bit multiply and clamps unsaturated tap coordinates: see the W role)
    if (a.i1.pitch != a.i2.pitch || (long long)a.H * a.i1.pitch * 8 > 0x7fffffffLL || a.i1.pitch * 8 >= (1 << 24) || a.H > 32766 || a.W > 32766) return LK_NOT_TAKEN;
    const int gx = split_column_blocks<C>(a);
    const long long pmax = std::max(a.u.pitch, a.v.pitch);
    const long long cap = (0x7fffffffLL / (pmax * 8) - 2 * C::R - 8) & ~1LL;
    if (cap < 2) return LKFLOW_E_ARG;
    const long long nblk = (long long)gx * pairs;
    StripPlan plan;
    if (ctx->tile_rows > 0) {
        const int th = (int)std::min<long long>((ctx->tile_rows + 1) & ~1, cap);
        plan = StripPlan{th, nblk, th};
    } else {
        const int bpc = blocks_per_cu(ctx, lk::lk_split_kernel<WIN, K, true, true>, C::NT);
        plan = cached_strip_plan(ctx, a.H, C::R + 2, nblk, (long long)ctx->n_cu * bpc, 0, cap);
        if (plan.thA > cap) plan = StripPlan{(int)cap, nblk, (int)cap};
    }
    const unsigned nmarch = apply_strip_plan(a, plan, gx, nblk);
    if (!nmarch) return LKFLOW_E_ARG;
    // the frame of clipped-window pixels and the outermost ring (its warped taps are made on the fly, lk_border_pixel<., true>):
    // trailing workgroups of the same launch, as in the unfused kernel
    const long long nr = std::min(2 * C::R, a.H), nc = std::min(2 * C::R, a.W);
    const long long nfp = nr * a.W + (a.H > 2 * C::R ? (long long)(a.H - 2 * C::R) * nc : 0);
    if (nfp + 2 * C::NT >= 0x7fffffffLL) return LKFLOW_E_ARG;
    const long long nbb = (nfp + C::NT - 1) / C::NT;
    dim3 grid(nmarch);
    if ((long long)grid.x + nbb * pairs > 0x7fffffffLL) return LKFLOW_E_ARG;
    a.nmain = grid.x; a.nbb = (unsigned)nbb;
    static const int no_border = knob("LKFLOW_ABL_NO_BORDER", 0);   // (timing-only ablation of a debug-knob build: results wrong)
    if (!no_border) grid.x += (unsigned)(nbb * pairs);
    ProfScope ps(ctx, LKFLOW_STAGE_LK, level, s);    // (opened only once the shape is taken: a fallback leaves no empty record)
    hipLaunchKernelGGL((lk::lk_split_kernel<WIN, K, true, true>), grid, dim3(C::NT), 0, s, a);
    HIP_TRY(ctx, hipGetLastError());
    return LKFLOW_OK;
}

#define LKFLOW_SPLIT_FUSED_MAX_WIN 31                // 12-wave workgroups, one per CU: <= 168 VGPRs
bool fused_split_applies(const lkflow_ctx *ctx, int win, int W, int H, int pairs);

int launch_splitw(lkflow_ctx *ctx, const lk::LKArgs &a, int pairs, int win, hipStream_t s, int level)
{
    switch (win) {
#define SP_CASE(WN) case WN: return launch_splitw_win<WN, 2>(ctx, a, pairs, s, level);
#ifdef LKFLOW_FAST_BUILD
        SP_CASE(15) SP_CASE(21) SP_CASE(31)
#ifdef LKFLOW_FAST_WINS_EXTRA
        SP_CASE(23) SP_CASE(25) SP_CASE(27) SP_CASE(29)
#endif
#else
        SP_CASE(5) SP_CASE(7) SP_CASE(9) SP_CASE(11) SP_CASE(13) SP_CASE(15) SP_CASE(17) SP_CASE(19) SP_CASE(21)
        SP_CASE(23) SP_CASE(25) SP_CASE(27) SP_CASE(29) SP_CASE(31)
#endif
#undef SP_CASE
        default: return LK_NOT_TAKEN;
    }
}

// shapes the wave-specialised kernel takes (the rest -- odd or tiny levels -- runs the one-role kernel)
// (odd sizes only without a coarser level -- the coarsest level of a pyramid is the only one that can be odd: 3840 x 2160 / L5 and
// 960 x 540 / L3 end in 240 x 135, where the one-role kernel took 0.044 ms at 21 x 21 and 0.12 ms at 25 x 25 for 8 / 32 pairs;
// 1080 x 1080 / L4 ends in 135 x 135)
bool split_applies(int win, int W, int H, bool has_up = true)
{
    static const int split = knob("LKFLOW_SPLIT", 1);            // A/B: 0 = one-role kernel
    const int R = win / 2;
    return split && win >= 5 && win <= LKFLOW_MAX_WIN && (!((W | H) & 1) || !has_up) && W >= 2 && H > R + LK_SPLIT_PF + 1;
}

// Windows 5..21 on even levels: run the warp of the level inside the wave-specialised LK kernel (lk_split.h, W role: the warped
// planes never reach memory)?  LKFLOW_OPT_FUSE_WARP 1: wherever the kernel takes the shape; 2 (default): where it was measured
// faster than warp + LK -- levels of at least 256 rows (a 12-wave workgroup has a CU to itself and marches a tall strip: its
// pipeline fill and its weight table are paid per strip; 32 x 1080p: level 0 1.09 -> 1.01 ms, level 1 0.319 -> 0.312; one pair:
// whole step 0.134 -> 0.122 ms; 8 x 3840x2160 at 21 x 21: 1.14 -> 1.07, at 31 x 31 -- where the unfused kernel runs one workgroup
// per CU as well -- 2.70 -> 2.29; the 245-frame 240 x 160 stream: 0.23 -> 0.27, hence not there: profiles/r04_a_fused_warp_ab.txt).
bool fused_split_applies(const lkflow_ctx *ctx, int win, int W, int H, int pairs)
{
    static const int env = knob("LKFLOW_SPLIT_FUSED", -1);       // A/B: overrides the option
    const int mode = env >= 0 ? env : ctx->fuse_warp;
    if (!mode || ctx->impl != 0 || win > LKFLOW_SPLIT_FUSED_MAX_WIN || !split_applies(win, W, H)) return false;
    if (mode == 1) return true;
    // Fused wins from 256 rows on at every window since the launch plan has two strip heights (pick_strip_plan: one workgroup per
    // CU made the fused kernel's last round expensive -- 32 x 960 x 540 at 15 x 15 was a wash, 0.308 fused against 0.299, and is
    // 0.258 against 0.307 now; 270-row levels: 0.071 against 0.088 at 5 x 5, 0.041 against 0.049 at 21 x 21); 160- and 80-row
    // levels lose when the launch fills the GPU (240 x 160 x 244 stream: 0.26 against 0.22, 0.117 against 0.083; 240 x 136 x 64:
    // 0.080 against 0.067) and win when it does not -- a launch of a few dozen workgroups is bound by its own latency, and one
    // launch beats two (16 x 320 x 224 at 7 x 7: 0.036 against 0.045; 160 x 112: 0.026 against 0.029; 80 x 56: 0.015 against
    // 0.027).  profiles/r04_g_two_heights_ab.txt
    return H >= 256 || (long long)pairs * W * H <= 1310720;
}

// LK_NOT_TAKEN: this window has no wave-specialised instantiation
int launch_split(lkflow_ctx *ctx, const lk::LKArgs &a, bool has_up, int pairs, int win, hipStream_t s)
{
    switch (win) {
#define SP_CASE(WN, KK) case WN: return launch_split_win<WN, KK>(ctx, a, has_up, pairs, s);
#ifdef LKFLOW_FAST_BUILD
#ifndef LK_SPLIT_K15
#define LK_SPLIT_K15 2
#endif
        SP_CASE(15, LK_SPLIT_K15) SP_CASE(21, 2) SP_CASE(31, 2)
#ifdef LKFLOW_FAST_WINS_EXTRA   /* experiments on the windows between */
        SP_CASE(23, 2) SP_CASE(25, 2) SP_CASE(27, 2) SP_CASE(29, 2)
#endif
#else
        SP_CASE(5, 2) SP_CASE(7, 2) SP_CASE(9, 2) SP_CASE(11, 2) SP_CASE(13, 2) SP_CASE(15, 2)
        SP_CASE(17, 2) SP_CASE(19, 2) SP_CASE(21, 2) SP_CASE(23, 2) SP_CASE(25, 2) SP_CASE(27, 2) SP_CASE(29, 2) SP_CASE(31, 2)
#endif
#undef SP_CASE
        default: return LK_NOT_TAKEN;
    }
}

#ifdef LKFLOW_EXPERIMENTAL
// Batch-synchronous kernel (lk_batch.h): 8 identical waves, 8 marched rows per phase pair; fw: a.i1 / a.i2 are the
// UN-warped level images and the warp of the level runs inside the kernel.  LK_NOT_TAKEN: shape not taken.
template <int WIN>
int launch_batch_win(lkflow_ctx *ctx, lk::LKArgs a, bool has_up, bool fw, int pairs, hipStream_t s, int stage, int level)
{
    using C = lk::BatchCfg<WIN>;
    if ((a.W & 1) || (a.H & 1) || a.W < 2 || a.H <= C::R + 3) return LK_NOT_TAKEN;
    if (a.i1.pitch != a.i2.pitch) return LK_NOT_TAKEN;           // (the kernel indexes both images with one pitch)
    if (fw && !has_up) return LKFLOW_E_ARG;
    const int gx = (a.W + C::XO - 1) / C::XO;
    int th = ctx->tile_rows;
    if (th <= 0) {
        const int bpc = fw ? blocks_per_cu(ctx, lk::lk_batch_kernel<WIN, true, true>, C::NT)
                           : (has_up ? blocks_per_cu(ctx, lk::lk_batch_kernel<WIN, true, false>, C::NT)
                                     : blocks_per_cu(ctx, lk::lk_batch_kernel<WIN, false, false>, C::NT));
        // a strip marches th + R + APR rows rounded up to whole batches, plus one more phase A
        th = pick_tile_rows(a.H, (C::R + C::APR + 12) / 2, (long long)gx * pairs, (long long)ctx->n_cu * bpc);
    }
    th = (th + 7) & ~7;                              // whole batches: the chunk position of a batch row is static
    a.th = th;
    a.gx = gx; a.gy = (a.H + th - 1) / th;
    if ((long long)a.gx * a.gy * pairs > 0x7fffffffLL) return LKFLOW_E_ARG;
    dim3 grid((unsigned)(a.gx * a.gy * pairs));      // 1-D: the kernel orders the tiles per XCD
    // the frame of clipped-window pixels (SerialPyr.cpp:94-95) and the outermost ring: trailing workgroups
    const long long nr = std::min(2 * C::R, a.H), nc = std::min(2 * C::R, a.W);
    const long long nfp = nr * a.W + (a.H > 2 * C::R ? (long long)(a.H - 2 * C::R) * nc : 0);
    if (nfp + 2 * C::NT >= 0x7fffffffLL) return LKFLOW_E_ARG;
    const long long nbb = (nfp + C::NT - 1) / C::NT;
    if ((long long)grid.x + nbb * pairs > 0x7fffffffLL) return LKFLOW_E_ARG;
    a.nmain = grid.x; a.nbb = (unsigned)nbb;
    grid.x += (unsigned)(nbb * pairs);
    ProfScope ps(ctx, stage, level, s);              // (opened only once the shape is taken)
    if (fw) hipLaunchKernelGGL((lk::lk_batch_kernel<WIN, true, true>), grid, dim3(C::NT), 0, s, a);
    else if (has_up) hipLaunchKernelGGL((lk::lk_batch_kernel<WIN, true, false>), grid, dim3(C::NT), 0, s, a);
    else hipLaunchKernelGGL((lk::lk_batch_kernel<WIN, false, false>), grid, dim3(C::NT), 0, s, a);
    return LKFLOW_OK;
}

// windows the batch-synchronous kernel is built for (<= 128 VGPRs with the WIN x 4 register ring); opt-in (LKFLOW_OPT_LK_BATCH)
bool batch_applies(const lkflow_ctx *ctx, int win, int W, int H)
{
    static const int env = knob("LKFLOW_BATCH", -1);   // A/B: overrides the option
    return (env >= 0 ? env : ctx->lk_batch) && ctx->impl == 0 && win >= 5 && win <= 15 && !(W & 1) && !(H & 1) && W >= 2 && H > win / 2 + 3;
}

int launch_batch(lkflow_ctx *ctx, const lk::LKArgs &a, bool has_up, bool fw, int pairs, int win, hipStream_t s, int stage, int level)
{
    switch (win) {
#define BT_CASE(WN) case WN: return launch_batch_win<WN>(ctx, a, has_up, fw, pairs, s, stage, level);
#ifdef LKFLOW_FAST_BUILD
        BT_CASE(15)
#else
        BT_CASE(5) BT_CASE(7) BT_CASE(9) BT_CASE(11) BT_CASE(13) BT_CASE(15)
#endif
#undef BT_CASE
        default: return LK_NOT_TAKEN;
    }
}

#else
inline bool batch_applies(const lkflow_ctx *, int, int, int) { return false; }
inline int launch_batch(lkflow_ctx *, const lk::LKArgs &, bool, bool, int, int, hipStream_t, int, int) { return LK_NOT_TAKEN; }
#endif

// 3 x 3 windows on even-width levels: two columns per lane (lk_w3x2.h)
int launch_w3x2(lkflow_ctx *ctx, lk::LKArgs a, bool has_up, int pairs, hipStream_t s)
{
    const int bpc = has_up ? blocks_per_cu(ctx, lk::lk_w3x2_kernel<true>, 64) : blocks_per_cu(ctx, lk::lk_w3x2_kernel<false>, 64);
    const int gx = (a.W + lk::W3X2_XO - 1) / lk::W3X2_XO;
    int th = ctx->tile_rows > 0 ? ctx->tile_rows : pick_tile_rows(a.H, 1, (long long)gx * pairs, (long long)ctx->n_cu * bpc);
    if (th > a.H) th = a.H;
    {
        const long long pmax = std::max(a.i1.pitch, std::max(a.u.pitch, a.v.pitch));
        const long long cap = 0x7fffffffLL / (pmax * 8) - 12;
        if (cap < 1) return LKFLOW_E_ARG;
        if (th > cap) th = (int)cap;
    }
    a.th = th;
    dim3 grid(gx, (a.H + th - 1) / th, pairs);
    if (has_up) hipLaunchKernelGGL(lk::lk_w3x2_kernel<true>, grid, dim3(64), 0, s, a);
    else hipLaunchKernelGGL(lk::lk_w3x2_kernel<false>, grid, dim3(64), 0, s, a);
    return LKFLOW_OK;
}

// warp + LK fused for 3 x 3 windows (lk_w3x2f.h): a.i1 / a.i2 are the UN-warped level images
bool fused_w3_applies(const lkflow_ctx *ctx, int win, int W, int H)
{
    static const bool off = (knob("LKFLOW_NO_W3_FUSED", 0) | knob("LKFLOW_NO_W3X2", 0)) != 0;
    return !off && ctx->impl == 0 && win == 3 && W >= 2 && !(W & 1) && !(H & 1);
}
#ifndef LKFLOW_W3_ONE_COLUMN
#define LKFLOW_W3_ONE_COLUMN 1                      /* fused 3 x 3: one column per lane with the gathers one row ahead (lk_w3x1f.h) instead of two columns */
#endif
int launch_w3x2f(lkflow_ctx *ctx, lk::LKArgs a, int pairs, hipStream_t s)
{
#ifdef LKFLOW_EXPERIMENTAL
    static const int one_col = knob("LKFLOW_W3_ONE_COLUMN", LKFLOW_W3_ONE_COLUMN);   // A/B
    const int xo = one_col ? lk::W3X1_XO : lk::W3X2_XO;
    const int bpc = one_col ? blocks_per_cu(ctx, lk::lk_w3x1f_kernel<true>, 64) : blocks_per_cu(ctx, lk::lk_w3x2f_kernel, 64);
#else
    constexpr int one_col = 1;
    const int xo = lk::W3X1_XO;
    const int bpc = blocks_per_cu(ctx, lk::lk_w3x1f_kernel<true>, 64);
#endif
    const int gx = (a.W + xo - 1) / xo;
    int th = ctx->tile_rows > 0 ? ctx->tile_rows : pick_tile_rows(a.H, 1, (long long)gx * pairs, (long long)ctx->n_cu * bpc);
    if (th > a.H) th = a.H;
    {
        const long long pmax = std::max(std::max(a.cu.pitch, a.cv.pitch), std::max(a.u.pitch, a.v.pitch));
        const long long cap = 0x7fffffffLL / (pmax * 8) - 12;
        if (cap < 1) return LKFLOW_E_ARG;
        if (th > cap) th = (int)cap;
    }
    a.th = th;
    if (one_col && a.i1.pitch < (1 << 24) && a.H < (1 << 24)) hipLaunchKernelGGL(lk::lk_w3x1f_kernel<true>, dim3(gx, (a.H + th - 1) / th, pairs), dim3(64), 0, s, a);
    else if (one_col) hipLaunchKernelGGL(lk::lk_w3x1f_kernel<false>, dim3(gx, (a.H + th - 1) / th, pairs), dim3(64), 0, s, a);
#ifdef LKFLOW_EXPERIMENTAL
    else hipLaunchKernelGGL(lk::lk_w3x2f_kernel, dim3(gx, (a.H + th - 1) / th, pairs), dim3(64), 0, s, a);
#endif
    HIP_TRY(ctx, hipGetLastError());
    return LKFLOW_OK;
}

// One-role kernel (lk_march3.h) for the shapes the wave-specialised kernel does not take: odd or tiny levels -- only ever
// the coarsest levels of a pyramid, where the time is negligible -- so every window is compiled for ONE block width:
// 128-column blocks (2 waves per barrier, ~10 % faster per thread-column than 256-column ones) up to 15 x 15, 256-column
// blocks above (the 2R halo columns would eat a 128-column block).  Results do not depend on the block width (the pair
// sums of a window are anchored to the image column's parity).  3 x 3 never comes here (launch_lk).
constexpr int lk3_block_width(int win) { return win <= 15 ? 128 : 256; }
int launch_lk3(lkflow_ctx *ctx, const lk::LKArgs &a, bool has_up, int pairs, int win, hipStream_t s)
{
    switch (win) {
#define LK_CASE(WN) case WN: return launch_lk3_win<WN, lk3_block_width(WN)>(ctx, a, has_up, pairs, s);
#ifdef LKFLOW_FAST_BUILD   /* A/B experiment builds: only the windows the goldens / headline use */
        LK_CASE(5) LK_CASE(9) LK_CASE(13) LK_CASE(15) LK_CASE(21)
#ifdef LKFLOW_FAST_WINS_EXTRA
        LK_CASE(23) LK_CASE(25) LK_CASE(27) LK_CASE(29) LK_CASE(31)
#endif
#else
        LK_CASE(5) LK_CASE(7) LK_CASE(9) LK_CASE(11) LK_CASE(13) LK_CASE(15) LK_CASE(17)
        LK_CASE(19) LK_CASE(21) LK_CASE(23) LK_CASE(25) LK_CASE(27) LK_CASE(29) LK_CASE(31)
#endif
#undef LK_CASE
        default: return LKFLOW_E_WIN;
    }
    return LKFLOW_OK;
}

int launch_lk(lkflow_ctx *ctx, lk::LKArgs a, bool has_up, int pairs, int win, hipStream_t s)
{
    if (ctx->impl == 1) {
        dim3 g = grid_px(a.W, a.H, pairs);
        if (has_up) hipLaunchKernelGGL((lk::lk_naive_kernel<true>), g, dim3(256), 0, s, a, win);
        else hipLaunchKernelGGL((lk::lk_naive_kernel<false>), g, dim3(256), 0, s, a, win);
        HIP_TRY(ctx, hipGetLastError());
        return LKFLOW_OK;
    }
    if (ctx->impl == 0) {
        int rc;
        static const bool no_x2 = knob("LKFLOW_NO_W3X2", 0) != 0;
        if (win == 3 && !no_x2 && a.W >= 2 && !(a.W & 1)) rc = launch_w3x2(ctx, a, has_up, pairs, s);   // 2 columns per lane
        else if (win == 3) rc = launch_lk3_win<3, 64>(ctx, a, has_up, pairs, s);   // barrier-free DPP kernel
        else {
            rc = batch_applies(ctx, win, a.W, a.H) ? launch_batch(ctx, a, has_up, false, pairs, win, s, -1, 0) : LK_NOT_TAKEN;
            if (rc == LK_NOT_TAKEN)
                rc = split_applies(win, a.W, a.H, has_up) ? launch_split(ctx, a, has_up, pairs, win, s) : LK_NOT_TAKEN;
            if (rc == LK_NOT_TAKEN) rc = launch_lk3(ctx, a, has_up, pairs, win, s);
        }
        if (rc) return rc;
        HIP_TRY(ctx, hipGetLastError());
        return LKFLOW_OK;
    }
    return LKFLOW_E_ARG;
}

int launch_pyrdown(lkflow_ctx *ctx, CPlanes src, int H, int W, Planes dst, int planes, hipStream_t s)
{
    lk::PyrDownArgs a{};
    a.src = src; a.dst = dst; a.H = H; a.W = W;
    a.Hd = (H + 1) / 2; a.Wd = (W + 1) / 2;
#ifndef LK_PD_TH
#define LK_PD_TH 16
#endif
    a.th = LK_PD_TH;
    a.gx = (a.Wd + 4 * lk::PD_WCOLS - 1) / (4 * lk::PD_WCOLS);
    // a launch that does not fill the GPU is as long as ONE workgroup's march (one 1080p pair: 272 workgroups of 16 rows 0.020 ms,
    // 1088 of 4 rows 0.013; profiles/r04_g_two_heights_ab.txt E): shorter strips until there are four workgroups per CU
    while (a.th > 4 && (long long)a.gx * ((a.Hd + a.th - 1) / a.th) * planes < 4LL * ctx->n_cu) a.th /= 2;
    { static const int e = knob("LKFLOW_PD_TH", 0); if (e > 0) a.th = e; }
    a.gy = (a.Hd + a.th - 1) / a.th;
    if ((long long)(2 * a.th + 6) * 8 * std::max(src.pitch, dst.pitch) > 0x7fffffffLL) return LKFLOW_E_ARG;   // strip-relative 32-bit offsets
    const long long n = (long long)a.gx * a.gy * planes;
    if (n > 0x7fffffffLL) return LKFLOW_E_ARG;
    hipLaunchKernelGGL(lk::pyrdown_kernel, dim3((unsigned)n), dim3(256), 0, s, a);
    HIP_TRY(ctx, hipGetLastError());
    return LKFLOW_OK;
}

// Levels 1 and 2 of a pyramid in one pass over level 0 (lk_kernels.h pyrdown2_kernel).  LK_NOT_TAKEN: shape not taken (the
// caller runs two single-level passes): both levels must halve exactly and be big enough for the in-wave reflections.
// Strip height: the kernel is latency-bound per wave (load -> filter -> DPP chain -> store), so MANY small workgroups beat tall
// strips although a strip of 8 level-1 rows computes 11 (32 x 2 x 1080p: 0.302 ms at 8 rows, 0.319 at 16 and 64, 0.333 at 32;
// two passes 0.335; profiles/r04_c_pyrdown2_ab.txt).  Small batches (the 240 x 160 stream: 0.054 against 0.055) keep two passes.
#ifndef LK_PD2_TH
#define LK_PD2_TH 8
#endif
bool pyrdown2_applies(int H, int W, int planes)
{
    static const int on = knob("LKFLOW_PD2", 1);              // A/B: 0 = two passes
    return on && H % 4 == 0 && W % 4 == 0 && W / 2 >= 8 && H / 2 >= 4 && (long long)(H / 2) * (W / 2) * planes >= (4ll << 20);
}
int launch_pyrdown2(lkflow_ctx *ctx, CPlanes src, int H, int W, Planes d1, Planes d2, int planes, hipStream_t s)
{
    if (H % 4 || W % 4 || W / 2 < 8 || H / 2 < 4) return LK_NOT_TAKEN;
    lk::PyrDown2Args a{};
    a.src = src; a.d1 = d1; a.d2 = d2; a.H = H; a.W = W;
    a.H1 = H / 2; a.W1 = W / 2; a.H2 = H / 4; a.W2 = W / 4;
    a.th = LK_PD2_TH;
    { static const int e = knob("LKFLOW_PD2_TH", 0); if (e > 0) a.th = (e + 3) & ~3; }
    a.gx = (a.W1 + 4 * lk::PD2_WCOLS - 1) / (4 * lk::PD2_WCOLS); a.gy = (a.H1 + a.th - 1) / a.th;
    if ((long long)(2 * a.th + 16) * 8 * std::max(src.pitch, std::max(d1.pitch, d2.pitch)) > 0x7fffffffLL) return LK_NOT_TAKEN;   // strip-relative 32-bit offsets
    const long long n = (long long)a.gx * a.gy * planes;
    if (n > 0x7fffffffLL) return LK_NOT_TAKEN;
    hipLaunchKernelGGL(lk::pyrdown2_kernel, dim3((unsigned)n), dim3(256), 0, s, a);
    HIP_TRY(ctx, hipGetLastError());
    return LKFLOW_OK;
}

int launch_warp(lkflow_ctx *ctx, CPlanes i1, CPlanes i2, int H, int W, CPlanes cu, CPlanes cv,
                Planes o1, Planes o2, int pairs, hipStream_t s)
{
    lk::WarpArgs a{};
    a.img1 = i1; a.img2 = i2; a.cu = cu; a.cv = cv; a.out1 = o1; a.out2 = o2;
    a.H = H; a.W = W;
#ifndef LK_WARP_TH
#define LK_WARP_TH 16
#endif
    a.th = LK_WARP_TH;                           // even; two rows per iteration
    { static const int e = knob("LKFLOW_WARP_TH", 0); if (e > 0) a.th = e & ~1; }
    a.gx = (W + 255) / 256; a.gy = (H + a.th - 1) / a.th;
    if ((long long)(a.th + 2) * 8 * std::max(std::max(o1.pitch, o2.pitch), std::max(cu.pitch, cv.pitch)) > 0x7fffffffLL)
        return LKFLOW_E_ARG;                      // strip-relative 32-bit byte offsets
    const long long n = (long long)a.gx * a.gy * pairs;
    if (n > 0x7fffffffLL) return LKFLOW_E_ARG;
    if (i1.pitch < (1 << 24) && H < (1 << 24)) hipLaunchKernelGGL(lk::warp_kernel<true>, dim3((unsigned)n), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(lk::warp_kernel<false>, dim3((unsigned)n), dim3(256), 0, s, a);
    HIP_TRY(ctx, hipGetLastError());
    return LKFLOW_OK;
}

int set_device(lkflow_ctx *ctx)
{
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return LKFLOW_OK;
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int lkflow_max_levels(int cols, int rows) { return max_levels(cols, rows); }

const char *lkflow_strerror(int status)
{
    switch (status) {
        case LKFLOW_OK: return "ok";
        case LKFLOW_E_ARG: return "invalid argument (null pointer, size, step or flags)";
        case LKFLOW_E_LEVELS: return "pyr_levels out of range [1, lkflow_max_levels(cols, rows)]";
        case LKFLOW_E_WIN: return "win must be odd, >= 3 and <= LKFLOW_MAX_WIN";
        case LKFLOW_E_HIP: return "HIP runtime error";
        case LKFLOW_E_NOMEM: return "device workspace allocation failed";
        case LKFLOW_E_NODEVICE: return "no usable HIP device";
        default: return "unknown lkflow status";
    }
}

const char *lkflow_last_error(const lkflow_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }
const char *lkflow_version(void) { return LKFLOW_VERSION_STR; }

int lkflow_lk_kernel_name(int rows, int cols, int win, int has_up, int fused_ok, int n_pairs, char *buf, size_t n)
{
    if (!buf || n == 0 || rows < 1 || cols < 1 || n_pairs < 1) return LKFLOW_E_ARG;
    int rc = check_win(win);
    if (rc) return rc;
    const char *up = has_up ? "true" : "false";
    static const bool no_fused = (knob("LKFLOW_NO_W3_FUSED", 0) | knob("LKFLOW_NO_W3X2", 0)) != 0;
    static const bool no_x2 = knob("LKFLOW_NO_W3X2", 0) != 0;
    if (win == 3 && has_up && fused_ok && !no_fused && cols >= 2 && !(cols & 1) && !(rows & 1))
        snprintf(buf, n, "%s", knob("LKFLOW_W3_ONE_COLUMN", LKFLOW_W3_ONE_COLUMN) ? "lk::lk_w3x1f_kernel<true>" : "lk::lk_w3x2f_kernel");
    else if (win == 3 && !no_x2 && cols >= 2 && !(cols & 1)) snprintf(buf, n, "lk::lk_w3x2_kernel<%s>", up);
    else if (win == 3) snprintf(buf, n, "lk::lk_march3_kernel<3, 64, %s>", up);
    else if (split_applies(win, cols, rows, has_up != 0)) {
        lkflow_ctx dflt;                                         // default options; the CU count of device 0 if there is one
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0) == hipSuccess && ncu > 0) dflt.n_cu = ncu;
        else (void)hipGetLastError();
        const bool fused = has_up && fused_ok && fused_split_applies(&dflt, win, cols, rows, n_pairs);
        snprintf(buf, n, "lk::lk_split_kernel<%d, 2, %s, %s>", win, up, fused ? "true" : "false");
    }
    else {
        snprintf(buf, n, "lk::lk_march3_kernel<%d, %d, %s>", win, lk3_block_width(win), up);
    }
    return LKFLOW_OK;
}

int lkflow_strip_plan(int rows, int halo_rows, long long column_strips, int slots, int lone_slots, long long out[3])
{
    if (!out || rows < 2 || halo_rows < 0 || column_strips < 1 || slots < 1 || lone_slots < 0) return LKFLOW_E_ARG;
    const StripPlan p = pick_strip_plan(rows, halo_rows, column_strips, slots, lone_slots, 0x7fffffff);
    out[0] = p.thA; out[1] = p.csA; out[2] = p.thB;
    return LKFLOW_OK;
}

int lkflow_ctx_create(lkflow_ctx **out, int device)
{
    if (!out) return LKFLOW_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n)
        return LKFLOW_E_NODEVICE;
    lkflow_ctx *ctx = new (std::nothrow) lkflow_ctx();
    if (!ctx) return LKFLOW_E_NOMEM;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return LKFLOW_E_HIP;
    }
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0)
        ctx->n_cu = ncu;
    *out = ctx;
    return LKFLOW_OK;
}

void lkflow_ctx_destroy(lkflow_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->stage) (void)hipFree(ctx->stage);
    for (auto &q : ctx->pins) unpin_host_range(q.p);
    for (auto &r : ctx->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int lkflow_ctx_set_option(lkflow_ctx *ctx, int key, int value)
{
    if (!ctx) return LKFLOW_E_ARG;
    switch (key) {
        case LKFLOW_OPT_IMPL:
            if (value < 0 || value > 1) return LKFLOW_E_ARG;
            ctx->impl = value;
            return LKFLOW_OK;
        case LKFLOW_OPT_FUSE_WARP:
            if (value < 0 || value > 2) return LKFLOW_E_ARG;
            ctx->fuse_warp = value;
            return LKFLOW_OK;
        case LKFLOW_OPT_LK_BATCH:
#ifdef LKFLOW_EXPERIMENTAL
            if (value < 0 || value > 1) return LKFLOW_E_ARG;
            ctx->lk_batch = value;
            return LKFLOW_OK;
#else
            return value == 0 ? LKFLOW_OK : LKFLOW_E_ARG;        // the kernel is not in this build (lkflow.h)
#endif
        case LKFLOW_OPT_TILE_ROWS:
            if (value < 0 || value > 4096) return LKFLOW_E_ARG;
            ctx->tile_rows = value;
            return LKFLOW_OK;
        case LKFLOW_OPT_PROFILE:
            if (value < -1 || value > LKFLOW_N_STAGES * LKFLOW_MAX_LEVELS) return LKFLOW_E_ARG;
            ctx->profile = value;
            return LKFLOW_OK;
        case LKFLOW_OPT_PIN_HOST:
            if (value < 0 || value > 1) return LKFLOW_E_ARG;
            if (!value && ctx->pin_host) {
                if (set_device(ctx)) return LKFLOW_E_HIP;
                (void)hipStreamSynchronize(ctx->own_stream);
                ctx->host_inflight = false;
                for (auto &q : ctx->pins) unpin_host_range(q.p);
                ctx->pins.clear();
            }
            ctx->pin_host = value;
            return LKFLOW_OK;
        default: return LKFLOW_E_ARG;
    }
}

size_t lkflow_ctx_workspace_bytes(const lkflow_ctx *ctx) { return ctx ? ctx->arena_bytes : 0; }

int lkflow_ctx_profile_read(lkflow_ctx *ctx, double *total_ms, int *n_launches)
{
    if (!ctx || !total_ms || !n_launches) return LKFLOW_E_ARG;
    for (int i = 0; i < LKFLOW_N_STAGES * LKFLOW_MAX_LEVELS; ++i) { total_ms[i] = 0.0; n_launches[i] = 0; }
    int rc = set_device(ctx);
    if (rc) return rc;
    std::vector<lkflow_ctx::Rec> recs;
    recs.swap(ctx->recs);                       // every record leaves the list exactly once, whatever happens below
    int status = LKFLOW_OK;
    for (auto &r : recs) {
        float ms = 0.f;
        if (status == LKFLOW_OK &&
            (!hip_ok(ctx, hipEventSynchronize(r.b), "hipEventSynchronize") ||
             !hip_ok(ctx, hipEventElapsedTime(&ms, r.a, r.b), "hipEventElapsedTime")))
            status = LKFLOW_E_HIP;
        if (status == LKFLOW_OK) { total_ms[r.slot] += ms; n_launches[r.slot] += 1; }
        ctx->ev_pool.push_back(r.a);
        ctx->ev_pool.push_back(r.b);
    }
    return status;
}

int lkflow_ctx_reserve(lkflow_ctx *ctx, int rows, int cols, int pyr_levels, int n_pairs)
{
    if (!ctx || rows < 1 || cols < 1 || n_pairs < 1 || n_pairs > 32767) return LKFLOW_E_ARG;
    if (pyr_levels < 1 || pyr_levels > max_levels(cols, rows)) return LKFLOW_E_LEVELS;
    int rc = set_device(ctx);
    if (rc) return rc;
    return ensure_workspace(ctx, rows, cols, pyr_levels, n_pairs);
}

// ---- single-level LK ---------------------------------------------------------------------------
int lkflow_lk_batch_f64_dev(lkflow_ctx *ctx, const double *img1, const double *img2,
                            size_t s1, size_t s2, int n_pairs, int rows, int cols, size_t step_bytes,
                            double *U, double *V, size_t uv_pair_stride_bytes, size_t uv_step_bytes,
                            int win, void *stream)
{
    if (!ctx || !img1 || !img2 || !U || !V) return LKFLOW_E_ARG;
    int rc = check_common(rows, cols, step_bytes, uv_step_bytes, n_pairs);
    if (rc) return rc;
    if (s1 % 8 || s2 % 8 || uv_pair_stride_bytes % 8) return LKFLOW_E_ARG;
    if ((rc = check_win(win))) return rc;
    if ((rc = set_device(ctx))) return rc;
    // both images must share one pair stride for the kernel's plane addressing
    lk::LKArgs a{};
    a.i1 = {img1, (long long)(s1 / 8), (long long)(step_bytes / 8)};
    a.i2 = {img2, (long long)(s2 / 8), (long long)(step_bytes / 8)};
    a.u = {U, (long long)(uv_pair_stride_bytes / 8), (long long)(uv_step_bytes / 8)};
    a.v = {V, (long long)(uv_pair_stride_bytes / 8), (long long)(uv_step_bytes / 8)};
    a.H = rows; a.W = cols;
    return launch_lk(ctx, a, false, n_pairs, win, (hipStream_t)stream);
}

int lkflow_lk_up_batch_f64_dev(lkflow_ctx *ctx, const double *img1, const double *img2, size_t s1, size_t s2, int n_pairs,
                               int rows, int cols, size_t step_bytes, const double *cu, const double *cv, size_t c_ps,
                               size_t c_step_bytes, double *U, double *V, size_t uv_pair_stride_bytes, size_t uv_step_bytes,
                               int win, void *stream)
{
    if (!ctx || !img1 || !img2 || !cu || !cv || !U || !V) return LKFLOW_E_ARG;
    int rc = check_common(rows, cols, step_bytes, uv_step_bytes, n_pairs);
    if (rc) return rc;
    if ((rows & 1) || (cols & 1)) return LKFLOW_E_ARG;       // a level below the coarsest is twice its coarser level
    if (s1 % 8 || s2 % 8 || uv_pair_stride_bytes % 8 || c_ps % 8 || c_step_bytes % 8 || c_step_bytes < (size_t)(cols / 2) * 8)
        return LKFLOW_E_ARG;
    if ((rc = check_win(win))) return rc;
    if ((rc = set_device(ctx))) return rc;
    lk::LKArgs a{};
    a.i1 = {img1, (long long)(s1 / 8), (long long)(step_bytes / 8)};
    a.i2 = {img2, (long long)(s2 / 8), (long long)(step_bytes / 8)};
    a.cu = {cu, (long long)(c_ps / 8), (long long)(c_step_bytes / 8)};
    a.cv = {cv, (long long)(c_ps / 8), (long long)(c_step_bytes / 8)};
    a.u = {U, (long long)(uv_pair_stride_bytes / 8), (long long)(uv_step_bytes / 8)};
    a.v = {V, (long long)(uv_pair_stride_bytes / 8), (long long)(uv_step_bytes / 8)};
    a.H = rows; a.W = cols;
    return launch_lk(ctx, a, true, n_pairs, win, (hipStream_t)stream);
}

// ---- full pyramid ------------------------------------------------------------------------------
int lkflow_compute_batch_f64_dev(lkflow_ctx *ctx, double *prev, double *next, size_t prev_ps,
                                 size_t next_ps, int n_pairs, int rows, int cols, size_t step_bytes,
                                 double *U, double *V, size_t uv_ps, size_t uv_step_bytes,
                                 int pyr_levels, int win, unsigned flags, void *stream)
{
    if (!ctx || !prev || !next || !U || !V) return LKFLOW_E_ARG;
    int rc = check_common(rows, cols, step_bytes, uv_step_bytes, n_pairs);
    if (rc) return rc;
    if (prev_ps % 8 || next_ps % 8 || uv_ps % 8) return LKFLOW_E_ARG;
    if (flags & ~LKFLOW_MUTATE_INPUTS) return LKFLOW_E_ARG;
    if ((flags & LKFLOW_MUTATE_INPUTS) && n_pairs != 1) return LKFLOW_E_ARG;
    if (pyr_levels < 1 || pyr_levels > max_levels(cols, rows)) return LKFLOW_E_LEVELS;
    if ((rc = check_win(win))) return rc;
    if ((rc = set_device(ctx))) return rc;
    hipStream_t s = (hipStream_t)stream;
    // An enqueued host-buffer pair (lkflow_enqueue_pair_*) works in this context's arena on the context's own stream:
    // a call on any OTHER stream must not carve or overwrite the arena under it (lkflow.h: one pair per context in flight).
    if (ctx->host_inflight && s != ctx->own_stream) {
        // (a host synchronisation is illegal while `s` is being captured into a graph: such a call is refused instead -- lkflow.h)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess) (void)hipGetLastError();
        if (cap != hipStreamCaptureStatusNone) { ctx->last_error = "capture while an enqueued host pair is unwaited: lkflow_wait() first"; return LKFLOW_E_ARG; }
        HIP_TRY(ctx, hipStreamSynchronize(ctx->own_stream));
        ctx->host_inflight = false;
    }
    if ((rc = ensure_workspace(ctx, rows, cols, pyr_levels, n_pairs))) return rc;

    const int L = pyr_levels, P = n_pairs;
    const std::vector<Level> &lv = ctx->lv;
    const long long step = (long long)(step_bytes / 8);

    // A video batch (pair p = frames p, p+1 of one buffer: next == prev + one frame stride, the addressing the
    // header documents) shares every interior frame between two pairs: its pyramid is built ONCE, P+1 planes per
    // level instead of 2P, and "frame 2" of pair p is simply plane p+1.  (The pyramids are never written after
    // they are built -- warps go to their own planes -- so sharing cannot change a result.)
    const bool video = P > 1 && prev_ps == next_ps && prev_ps >= (size_t)rows * step_bytes &&
                       reinterpret_cast<const char *>(next) == reinterpret_cast<const char *>(prev) + prev_ps;
    const long long f2 = video ? 1 : P;          // plane offset of a pair's second frame inside a level

    // image planes of frame f (0 = prev, 1 = next) at level l, original (un-warped)
    auto pyr = [&](int l, int f) -> CPlanes {
        if (l == 0) return f == 0 ? CPlanes{prev, (long long)(prev_ps / 8), step}
                                  : CPlanes{next, (long long)(next_ps / 8), step};
        return CPlanes{arena_at(ctx, ctx->off_pyr[l] + (long long)f * f2 * lv[l].plane), lv[l].plane,
                       lv[l].pitch};
    };
    auto pyr_w = [&](int l, int f) -> Planes {
        return Planes{arena_at(ctx, ctx->off_pyr[l] + (long long)f * f2 * lv[l].plane), lv[l].plane,
                      lv[l].pitch};
    };
    auto warp_w = [&](int l, int f) -> Planes {
        return Planes{arena_at(ctx, ctx->off_warp[l] + (long long)f * P * lv[l].plane), lv[l].plane,
                      lv[l].pitch};
    };
    auto flow_w = [&](int l, int c) -> Planes {  // c: 0 = u, 1 = v
        if (l == 0) return c == 0 ? Planes{U, (long long)(uv_ps / 8), (long long)(uv_step_bytes / 8)}
                                  : Planes{V, (long long)(uv_ps / 8), (long long)(uv_step_bytes / 8)};
        return Planes{arena_at(ctx, ctx->off_flow[l] + (long long)c * P * lv[l].plane), lv[l].plane,
                      lv[l].pitch};
    };
    auto as_c = [](Planes p) { return CPlanes{p.base, p.plane, p.pitch}; };

    // a2: Gaussian pyramids of both frames (built from the ORIGINAL images, before any warp).  Levels 1 and 2 come out of ONE
    // pass over level 0 where the shape allows (pyrdown2_kernel: level 1 is never read back)
    int l_first = 1;
    if (L >= 3 && pyrdown2_applies(lv[0].H, lv[0].W, video ? P + 1 : 2 * P)) {
        ProfScope ps(ctx, LKFLOW_STAGE_PYRDOWN, 1, s);          // (levels 1 + 2: there is no separate level-2 record)
        int rc2 = LKFLOW_OK;
        if (video) rc2 = launch_pyrdown2(ctx, pyr(0, 0), lv[0].H, lv[0].W, pyr_w(1, 0), pyr_w(2, 0), P + 1, s);
        else
            for (int f = 0; f < 2 && rc2 == LKFLOW_OK; ++f)
                rc2 = launch_pyrdown2(ctx, pyr(0, f), lv[0].H, lv[0].W, pyr_w(1, f), pyr_w(2, f), P, s);
        if (rc2 == LKFLOW_OK) l_first = 3;
        else if (rc2 != LK_NOT_TAKEN) return rc2;               // (not taken -- offsets too large for the strip -- : two passes below)
    }
    for (int l = l_first; l < L; ++l) {
        if (l == 1) {
            ProfScope ps(ctx, LKFLOW_STAGE_PYRDOWN, 1, s);
            if (video) {                                  // P+1 frames, one launch
                if ((rc = launch_pyrdown(ctx, pyr(0, 0), lv[0].H, lv[0].W, pyr_w(1, 0), P + 1, s))) return rc;
            } else {
                for (int f = 0; f < 2; ++f)
                    if ((rc = launch_pyrdown(ctx, pyr(0, f), lv[0].H, lv[0].W, pyr_w(1, f), P, s))) return rc;
            }
        } else {
            ProfScope ps(ctx, LKFLOW_STAGE_PYRDOWN, l, s);
            // the planes of a level are adjacent in the arena: one launch over 2P (video: P+1) planes
            if ((rc = launch_pyrdown(ctx, pyr(l - 1, 0), lv[l - 1].H, lv[l - 1].W, pyr_w(l, 0),
                                     video ? P + 1 : 2 * P, s)))
                return rc;
        }
    }

    // a13: coarse-to-fine loop
    for (int l = L - 1; l >= 0; --l) {
        lk::LKArgs a{};
        a.H = lv[l].H; a.W = lv[l].W;
        a.u = flow_w(l, 0); a.v = flow_w(l, 1);
        const bool has_up = l != L - 1;
        if (has_up && fused_w3_applies(ctx, win, lv[l].W, lv[l].H) && !((flags & LKFLOW_MUTATE_INPUTS) && l == 0)) {
            // 3 x 3: warp and LK in one kernel, the warped images never reach memory (lk_w3x2f.h)
            a.i1 = pyr(l, 0); a.i2 = pyr(l, 1);
            a.cu = as_c(flow_w(l + 1, 0)); a.cv = as_c(flow_w(l + 1, 1));
            ProfScope ps(ctx, LKFLOW_STAGE_LK, l, s);
            if ((rc = launch_w3x2f(ctx, a, P, s))) return rc;
            continue;
        }
        if (has_up && ctx->fuse_warp == 1 && batch_applies(ctx, win, lv[l].W, lv[l].H) && !((flags & LKFLOW_MUTATE_INPUTS) && l == 0)) {
            // windows 5..15: warp and LK in one batch-synchronous kernel (lk_batch.h), the warped images never reach memory
            lk::LKArgs f = a;
            f.i1 = pyr(l, 0); f.i2 = pyr(l, 1);
            f.cu = as_c(flow_w(l + 1, 0)); f.cv = as_c(flow_w(l + 1, 1));
            rc = launch_batch(ctx, f, true, true, P, win, s, LKFLOW_STAGE_LK, l);
            if (rc == LKFLOW_OK) { HIP_TRY(ctx, hipGetLastError()); continue; }
            if (rc != LK_NOT_TAKEN) return rc;                   // not taken: warp + LK below
        } else
        if (has_up && fused_split_applies(ctx, win, lv[l].W, lv[l].H, P) && !((flags & LKFLOW_MUTATE_INPUTS) && l == 0)) {
            // windows 5..15: warp and LK in one kernel (lk_split.h, W role), the warped images never reach memory
            lk::LKArgs f = a;
            f.i1 = pyr(l, 0); f.i2 = pyr(l, 1);
            f.cu = as_c(flow_w(l + 1, 0)); f.cv = as_c(flow_w(l + 1, 1));
            rc = launch_splitw(ctx, f, P, win, s, l);
            if (rc == LKFLOW_OK) continue;
            if (rc != LK_NOT_TAKEN) return rc;                   // not taken: warp + LK below
        }
        if (has_up) {
            CPlanes cu = as_c(flow_w(l + 1, 0)), cv = as_c(flow_w(l + 1, 1));
            // a9-a12: warp level l with the up-sampled flow of level l+1
            ProfScope ps(ctx, LKFLOW_STAGE_WARP, l, s);
            if ((rc = launch_warp(ctx, pyr(l, 0), pyr(l, 1), lv[l].H, lv[l].W, cu, cv, warp_w(l, 0),
                                  warp_w(l, 1), P, s)))
                return rc;
            a.i1 = as_c(warp_w(l, 0)); a.i2 = as_c(warp_w(l, 1));
            a.cu = cu; a.cv = cv;
        } else {
            a.i1 = pyr(l, 0); a.i2 = pyr(l, 1);
        }
        {
            ProfScope ps(ctx, LKFLOW_STAGE_LK, l, s);
            if ((rc = launch_lk(ctx, a, has_up, P, win, s))) return rc;
        }
    }

    // a12 at level 0: the reference's pyr[0] IS the caller's buffer, so its warp lands there
    if ((flags & LKFLOW_MUTATE_INPUTS) && L >= 2) {
        Planes w1 = warp_w(0, 0), w2 = warp_w(0, 1);
        HIP_TRY(ctx, hipMemcpy2DAsync(prev, step_bytes, w1.base, (size_t)w1.pitch * 8, (size_t)cols * 8,
                                      rows, hipMemcpyDeviceToDevice, s));
        HIP_TRY(ctx, hipMemcpy2DAsync(next, step_bytes, w2.base, (size_t)w2.pitch * 8, (size_t)cols * 8,
                                      rows, hipMemcpyDeviceToDevice, s));
    }
    return LKFLOW_OK;
}

// ---- host-buffer drop-ins ----------------------------------------------------------------------
static int ensure_stage(lkflow_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->stage_bytes) return LKFLOW_OK;
    HIP_TRY(ctx, hipDeviceSynchronize());
    if (ctx->stage) { HIP_TRY(ctx, hipFree(ctx->stage)); ctx->stage = nullptr; ctx->stage_bytes = 0; }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) { hip_ok(ctx, e, "hipMalloc(stage)"); return LKFLOW_E_NOMEM; }
    ctx->stage = (double *)p;
    ctx->stage_bytes = bytes;
    return LKFLOW_OK;
}

// LKFLOW_OPT_PIN_HOST: page-lock [p, p + n) once (kept in the context); failure is not an error -- the copy then
// takes the runtime's pageable path, as without the option.
static void pin_host_range(lkflow_ctx *ctx, const void *ptr, size_t n)
{
    if (!ctx->pin_host || !ptr || !n) return;
    char *p = (char *)const_cast<void *>(ptr);
    for (auto &q : ctx->pins)
        if (p >= q.p && p + n <= q.p + q.n) return;
    if (ctx->pins.size() >= 16) {                            // oldest out
        unpin_host_range(ctx->pins.front().p);
        ctx->pins.erase(ctx->pins.begin());
    }
    if (hipHostRegister(p, n, hipHostRegisterDefault) == hipSuccess) ctx->pins.push_back({p, n});
    else (void)hipGetLastError();
}

// H2D (+ convertTo on the device for 8-bit frames), the pyramid, D2H: all enqueued on the context's own stream.
static int enqueue_host_pair(lkflow_ctx *ctx, const void *prev, const void *next, bool u8, int rows, int cols,
                             size_t step_bytes, double *U, double *V, size_t uv_step_bytes, int pyr_levels, int win,
                             unsigned flags)
{
    if (!ctx || !prev || !next || !U || !V) return LKFLOW_E_ARG;
    const size_t esz = u8 ? 1 : 8;
    if (rows < 1 || cols < 1 || step_bytes < (size_t)cols * esz || step_bytes % esz) return LKFLOW_E_ARG;
    int rc = check_common(rows, cols, (size_t)cols * 8, uv_step_bytes, 1);
    if (rc) return rc;
    if (!u8 && (rc = check_common(rows, cols, step_bytes, uv_step_bytes, 1))) return rc;
    if (flags & ~LKFLOW_MUTATE_INPUTS) return LKFLOW_E_ARG;
    if (u8 && flags) return LKFLOW_E_ARG;                    // 8-bit frames cannot hold the warped CV_64F images
    if (pyr_levels < 1 || pyr_levels > max_levels(cols, rows)) return LKFLOW_E_LEVELS;
    if ((rc = check_win(win))) return rc;
    if ((rc = set_device(ctx))) return rc;
    hipStream_t s = ctx->own_stream;
    if (ctx->host_inflight) { HIP_TRY(ctx, hipStreamSynchronize(s)); ctx->host_inflight = false; }   // one pair per context in flight
    const size_t npx = (size_t)rows * cols, row_b = (size_t)cols * 8, plane_b = row_b * rows;
    if ((rc = ensure_stage(ctx, 4 * plane_b + (u8 ? 2 * npx + 64 : 0)))) return rc;
    double *d_prev = ctx->stage, *d_next = d_prev + npx, *d_u = d_next + npx, *d_v = d_u + npx;
    pin_host_range(ctx, prev, (size_t)(rows - 1) * step_bytes + (size_t)cols * esz);
    pin_host_range(ctx, next, (size_t)(rows - 1) * step_bytes + (size_t)cols * esz);
    pin_host_range(ctx, U, (size_t)(rows - 1) * uv_step_bytes + row_b);
    pin_host_range(ctx, V, (size_t)(rows - 1) * uv_step_bytes + row_b);
    // from here on copies may be in flight on the caller's buffers: any failure drains the stream before returning
    struct Drain { lkflow_ctx *c; hipStream_t st; bool armed = true; ~Drain() { if (armed) { (void)hipStreamSynchronize(st); c->host_inflight = false; } } } drain{ctx, s};
    ctx->host_inflight = true;
    if (u8) {
        unsigned char *d8 = reinterpret_cast<unsigned char *>(d_v + npx);       // [prev | next], contiguous
        HIP_TRY(ctx, hipMemcpy2DAsync(d8, cols, prev, step_bytes, cols, rows, hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemcpy2DAsync(d8 + npx, cols, next, step_bytes, cols, rows, hipMemcpyHostToDevice, s));
        if ((rc = lkflow_u8_to_f64_dev(ctx, d8, d_prev, 2 * npx, s))) return rc; // SerialPyr.cpp:237 on the device
    } else {
        HIP_TRY(ctx, hipMemcpy2DAsync(d_prev, row_b, prev, step_bytes, row_b, rows, hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemcpy2DAsync(d_next, row_b, next, step_bytes, row_b, rows, hipMemcpyHostToDevice, s));
    }
    rc = lkflow_compute_batch_f64_dev(ctx, d_prev, d_next, plane_b, plane_b, 1, rows, cols, row_b, d_u, d_v,
                                      plane_b, row_b, pyr_levels, win, flags, s);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpy2DAsync(U, uv_step_bytes, d_u, row_b, row_b, rows, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpy2DAsync(V, uv_step_bytes, d_v, row_b, row_b, rows, hipMemcpyDeviceToHost, s));
    if ((flags & LKFLOW_MUTATE_INPUTS) && pyr_levels >= 2) {
        HIP_TRY(ctx, hipMemcpy2DAsync(const_cast<void *>(prev), step_bytes, d_prev, row_b, row_b, rows, hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipMemcpy2DAsync(const_cast<void *>(next), step_bytes, d_next, row_b, row_b, rows, hipMemcpyDeviceToHost, s));
    }
    drain.armed = false;
    return LKFLOW_OK;
}

int lkflow_wait(lkflow_ctx *ctx)
{
    if (!ctx) return LKFLOW_E_ARG;
    int rc = set_device(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->own_stream));
    ctx->host_inflight = false;
    return LKFLOW_OK;
}

int lkflow_enqueue_pair_f64(lkflow_ctx *ctx, double *prev, double *next, int rows, int cols, size_t step_bytes,
                            double *U, double *V, size_t uv_step_bytes, int pyr_levels, int win, unsigned flags)
{
    return enqueue_host_pair(ctx, prev, next, false, rows, cols, step_bytes, U, V, uv_step_bytes, pyr_levels, win, flags);
}

int lkflow_enqueue_pair_u8(lkflow_ctx *ctx, const unsigned char *prev, const unsigned char *next, int rows, int cols,
                           size_t step_bytes, double *U, double *V, size_t uv_step_bytes, int pyr_levels, int win)
{
    return enqueue_host_pair(ctx, prev, next, true, rows, cols, step_bytes, U, V, uv_step_bytes, pyr_levels, win, 0);
}

int lkflow_compute_f64(lkflow_ctx *ctx, double *prev, double *next, int rows, int cols,
                       size_t step_bytes, double *U, double *V, size_t uv_step_bytes, int pyr_levels,
                       int win, unsigned flags)
{
    int rc = enqueue_host_pair(ctx, prev, next, false, rows, cols, step_bytes, U, V, uv_step_bytes, pyr_levels, win, flags);
    return rc ? rc : lkflow_wait(ctx);
}

int lkflow_compute_u8(lkflow_ctx *ctx, const unsigned char *prev, const unsigned char *next, int rows, int cols,
                      size_t step_bytes, double *U, double *V, size_t uv_step_bytes, int pyr_levels, int win)
{
    int rc = enqueue_host_pair(ctx, prev, next, true, rows, cols, step_bytes, U, V, uv_step_bytes, pyr_levels, win, 0);
    return rc ? rc : lkflow_wait(ctx);
}

int lkflow_lk_f64(lkflow_ctx *ctx, const double *img1, const double *img2, int rows, int cols,
                  size_t step_bytes, double *U, double *V, size_t uv_step_bytes, int win)
{
    if (!ctx || !img1 || !img2 || !U || !V) return LKFLOW_E_ARG;
    int rc = check_common(rows, cols, step_bytes, uv_step_bytes, 1);
    if (rc) return rc;
    if ((rc = check_win(win))) return rc;
    if ((rc = lkflow_wait(ctx))) return rc;                  // the staging planes may belong to an enqueued pair
    const size_t row_b = (size_t)cols * 8, plane_b = row_b * rows;
    if ((rc = ensure_stage(ctx, 4 * plane_b))) return rc;
    double *d_1 = ctx->stage, *d_2 = d_1 + (size_t)rows * cols, *d_u = d_2 + (size_t)rows * cols,
           *d_v = d_u + (size_t)rows * cols;
    hipStream_t s = ctx->own_stream;
    HIP_TRY(ctx, hipMemcpy2DAsync(d_1, row_b, img1, step_bytes, row_b, rows, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpy2DAsync(d_2, row_b, img2, step_bytes, row_b, rows, hipMemcpyHostToDevice, s));
    rc = lkflow_lk_batch_f64_dev(ctx, d_1, d_2, plane_b, plane_b, 1, rows, cols, row_b, d_u, d_v, plane_b,
                                 row_b, win, s);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpy2DAsync(U, uv_step_bytes, d_u, row_b, row_b, rows, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpy2DAsync(V, uv_step_bytes, d_v, row_b, row_b, rows, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return LKFLOW_OK;
}

// ---- stage entry points ------------------------------------------------------------------------
int lkflow_pyrdown_f64_dev(lkflow_ctx *ctx, const double *src, size_t src_ps, int n_planes, int rows,
                           int cols, size_t src_step_bytes, double *dst, size_t dst_ps,
                           size_t dst_step_bytes, void *stream)
{
    if (!ctx || !src || !dst || rows < 1 || cols < 1 || n_planes < 1 || n_planes > 65535) return LKFLOW_E_ARG;
    if (src_step_bytes < (size_t)cols * 8 || src_step_bytes % 8 || src_ps % 8 || dst_ps % 8) return LKFLOW_E_ARG;
    if (dst_step_bytes < (size_t)((cols + 1) / 2) * 8 || dst_step_bytes % 8) return LKFLOW_E_ARG;
    int rc = set_device(ctx);
    if (rc) return rc;
    return launch_pyrdown(ctx, CPlanes{src, (long long)(src_ps / 8), (long long)(src_step_bytes / 8)}, rows,
                          cols, Planes{dst, (long long)(dst_ps / 8), (long long)(dst_step_bytes / 8)},
                          n_planes, (hipStream_t)stream);
}

int lkflow_pyrdown2_f64_dev(lkflow_ctx *ctx, const double *src, size_t src_ps, int n_planes, int rows, int cols,
                            size_t src_step_bytes, double *dst1, size_t dst1_ps, size_t dst1_step_bytes, double *dst2,
                            size_t dst2_ps, size_t dst2_step_bytes, void *stream)
{
    if (!ctx || !src || !dst1 || !dst2 || rows < 1 || cols < 1 || n_planes < 1 || n_planes > 65535) return LKFLOW_E_ARG;
    if (src_step_bytes < (size_t)cols * 8 || src_step_bytes % 8 || src_ps % 8 || dst1_ps % 8 || dst2_ps % 8) return LKFLOW_E_ARG;
    if (rows % 4 || cols % 4 || cols / 2 < 8 || rows / 2 < 4) return LKFLOW_E_ARG;            // (the shapes the fused pass takes)
    if (dst1_step_bytes < (size_t)(cols / 2) * 8 || dst1_step_bytes % 8 || dst2_step_bytes < (size_t)(cols / 4) * 8 || dst2_step_bytes % 8)
        return LKFLOW_E_ARG;
    int rc = set_device(ctx);
    if (rc) return rc;
    rc = launch_pyrdown2(ctx, CPlanes{src, (long long)(src_ps / 8), (long long)(src_step_bytes / 8)}, rows, cols,
                         Planes{dst1, (long long)(dst1_ps / 8), (long long)(dst1_step_bytes / 8)},
                         Planes{dst2, (long long)(dst2_ps / 8), (long long)(dst2_step_bytes / 8)}, n_planes, (hipStream_t)stream);
    return rc == LK_NOT_TAKEN ? LKFLOW_E_ARG : rc;
}

int lkflow_pyrup_f64_dev(lkflow_ctx *ctx, const double *src, size_t src_ps, int n_planes, int rows,
                         int cols, size_t src_step_bytes, double *dst, size_t dst_ps,
                         size_t dst_step_bytes, void *stream)
{
    if (!ctx || !src || !dst || rows < 1 || cols < 1 || n_planes < 1 || n_planes > 65535) return LKFLOW_E_ARG;
    if (src_step_bytes < (size_t)cols * 8 || src_step_bytes % 8 || src_ps % 8 || dst_ps % 8) return LKFLOW_E_ARG;
    if (dst_step_bytes < (size_t)cols * 16 || dst_step_bytes % 8) return LKFLOW_E_ARG;
    int rc = set_device(ctx);
    if (rc) return rc;
    hipLaunchKernelGGL(lk::pyrup_kernel, grid_px(2 * cols, 2 * rows, n_planes), dim3(256), 0,
                       (hipStream_t)stream,
                       CPlanes{src, (long long)(src_ps / 8), (long long)(src_step_bytes / 8)}, rows, cols,
                       Planes{dst, (long long)(dst_ps / 8), (long long)(dst_step_bytes / 8)});
    HIP_TRY(ctx, hipGetLastError());
    return LKFLOW_OK;
}

int lkflow_warp_f64_dev(lkflow_ctx *ctx, const double *img1, const double *img2, size_t img_ps, int n_pairs,
                        int rows, int cols, size_t img_step_bytes, const double *cu, const double *cv,
                        size_t c_ps, size_t c_step_bytes, double *out1, double *out2, size_t out_ps,
                        size_t out_step_bytes, void *stream)
{
    if (!ctx || !img1 || !img2 || !cu || !cv || !out1 || !out2) return LKFLOW_E_ARG;
    if (rows < 2 || cols < 2 || (rows & 1) || (cols & 1) || n_pairs < 1 || n_pairs > 65535) return LKFLOW_E_ARG;
    if (img_step_bytes < (size_t)cols * 8 || out_step_bytes < (size_t)cols * 8 ||
        c_step_bytes < (size_t)(cols / 2) * 8)
        return LKFLOW_E_ARG;
    if (img_step_bytes % 8 || out_step_bytes % 8 || c_step_bytes % 8 || img_ps % 8 || c_ps % 8 || out_ps % 8)
        return LKFLOW_E_ARG;
    if (out1 == img1 || out2 == img2 || out1 == img2 || out2 == img1) return LKFLOW_E_ARG;
    int rc = set_device(ctx);
    if (rc) return rc;
    const long long ip = img_ps / 8, is = img_step_bytes / 8, cp = c_ps / 8, cs = c_step_bytes / 8,
                    op = out_ps / 8, os = out_step_bytes / 8;
    return launch_warp(ctx, CPlanes{img1, ip, is}, CPlanes{img2, ip, is}, rows, cols, CPlanes{cu, cp, cs},
                       CPlanes{cv, cp, cs}, Planes{out1, op, os}, Planes{out2, op, os}, n_pairs,
                       (hipStream_t)stream);
}

int lkflow_solve_f64_dev(lkflow_ctx *ctx, const double *S, size_t n, double *u, double *v, int plain_divisions, void *stream)
{
    if (!ctx || !S || !u || !v || plain_divisions < 0 || plain_divisions > 1) return LKFLOW_E_ARG;
    if (n == 0) return LKFLOW_OK;
    int rc = set_device(ctx);
    if (rc) return rc;
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(lk::solve_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, S, n, u, v, plain_divisions);
    HIP_TRY(ctx, hipGetLastError());
    return LKFLOW_OK;
}

int lkflow_u8_to_f64_dev(lkflow_ctx *ctx, const unsigned char *src, double *dst, size_t n, void *stream)
{
    if (!ctx || !src || !dst) return LKFLOW_E_ARG;
    if (n == 0) return LKFLOW_OK;
    int rc = set_device(ctx);
    if (rc) return rc;
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(lk::u8_to_f64_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src,
                       dst, n);
    HIP_TRY(ctx, hipGetLastError());
    return LKFLOW_OK;
}

static int edge_map_launch(lkflow_ctx *ctx, const unsigned char *src, const unsigned char *prev, double rate,
                           size_t src_ps, int n_planes, int rows, int cols, size_t src_step_bytes, int thresh,
                           double *dst, size_t dst_ps, size_t dst_step_bytes, void *stream)
{
    if (!ctx || !src || !dst || rows < 1 || cols < 1 || n_planes < 1) return LKFLOW_E_ARG;
    if (src_step_bytes < (size_t)cols || dst_step_bytes < (size_t)cols * 8 || dst_step_bytes % 8 || dst_ps % 8)
        return LKFLOW_E_ARG;
    int rc = set_device(ctx);
    if (rc) return rc;
    lk::EdgeArgs a{};
    a.src = src; a.prev = prev; a.beta = (float)(-rate);
    a.src_plane = (long long)src_ps; a.src_pitch = (long long)src_step_bytes;
    a.dst = Planes{dst, (long long)(dst_ps / 8), (long long)(dst_step_bytes / 8)};
    a.H = rows; a.W = cols; a.thresh = thresh;
    a.gx = (cols + lk::EM_TW - 1) / lk::EM_TW; a.gy = (rows + lk::EM_TH - 1) / lk::EM_TH;
    const long long n = (long long)a.gx * a.gy * n_planes;
    if (n > 0x7fffffffLL) return LKFLOW_E_ARG;
    hipLaunchKernelGGL(lk::edge_map_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(ctx, hipGetLastError());
    return LKFLOW_OK;
}

int lkflow_edge_map_u8_dev(lkflow_ctx *ctx, const unsigned char *src, size_t src_ps, int n_planes, int rows,
                           int cols, size_t src_step_bytes, int thresh, double *dst, size_t dst_ps,
                           size_t dst_step_bytes, void *stream)
{
    return edge_map_launch(ctx, src, nullptr, 0.0, src_ps, n_planes, rows, cols, src_step_bytes, thresh, dst, dst_ps,
                           dst_step_bytes, stream);
}

int lkflow_diff_edge_map_u8_dev(lkflow_ctx *ctx, const unsigned char *cur, const unsigned char *prev, size_t src_ps,
                                int n_planes, int rows, int cols, size_t src_step_bytes, double rate, int thresh,
                                double *dst, size_t dst_ps, size_t dst_step_bytes, void *stream)
{
    if (!prev) return LKFLOW_E_ARG;
    return edge_map_launch(ctx, cur, prev, rate, src_ps, n_planes, rows, cols, src_step_bytes, thresh, dst, dst_ps,
                           dst_step_bytes, stream);
}

int lkflow_blur_gray_bgr8_dev(lkflow_ctx *ctx, const unsigned char *bgr, size_t src_ps, int n_frames, int rows,
                              int cols, size_t src_step_bytes, int ksize, double sigma, unsigned char *gray,
                              size_t dst_ps, size_t dst_step_bytes, void *stream)
{
    if (!ctx || !bgr || !gray || rows < 1 || cols < 1 || n_frames < 1) return LKFLOW_E_ARG;
    if (src_step_bytes < (size_t)cols * 3 || dst_step_bytes < (size_t)cols) return LKFLOW_E_ARG;
    if (ksize < 1 || ksize > 2 * lk::BG_MAXR + 1 || !(ksize & 1)) return LKFLOW_E_WIN;
    if (!(sigma > 0.0)) return LKFLOW_E_ARG;
    int rc = set_device(ctx);
    if (rc) return rc;
    lk::BlurGrayArgs a{};
    // cv::getGaussianKernel(ksize, sigma, CV_32F) -> convertTo(CV_32S, 256) (OpenCV 2.4 smooth.cpp / filter.cpp)
    {
        float cf[2 * lk::BG_MAXR + 1];
        const double scale2x = -0.5 / (sigma * sigma);
        double sum = 0.0;
        for (int i = 0; i < ksize; ++i) {
            const double x = i - (ksize - 1) * 0.5;
            cf[i] = (float)std::exp(scale2x * x * x);
            sum += cf[i];
        }
        sum = 1.0 / sum;
        for (int i = 0; i < ksize; ++i) {
            cf[i] = (float)(cf[i] * sum);
            a.q[i] = (int)std::lrint((double)(cf[i] * 256.0f));
        }
    }
    a.src = bgr; a.src_plane = (long long)src_ps; a.src_pitch = (long long)src_step_bytes;
    a.dst = gray; a.dst_plane = (long long)dst_ps; a.dst_pitch = (long long)dst_step_bytes;
    a.H = rows; a.W = cols;
    a.gx = (cols + lk::BG_TW - 1) / lk::BG_TW; a.gy = (rows + lk::BG_TH - 1) / lk::BG_TH;
    const long long n = (long long)a.gx * a.gy * n_frames;
    if (n > 0x7fffffffLL) return LKFLOW_E_ARG;
    switch (ksize / 2) {
#define BG_CASE(RR) case RR: hipLaunchKernelGGL(lk::blur_gray_kernel<RR>, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, a); break;
        BG_CASE(0) BG_CASE(1) BG_CASE(2) BG_CASE(3) BG_CASE(4) BG_CASE(5) BG_CASE(6) BG_CASE(7)
#undef BG_CASE
        default: return LKFLOW_E_WIN;
    }
    HIP_TRY(ctx, hipGetLastError());
    return LKFLOW_OK;
}

int lkflow_resize_cubic_bgr8_dev(lkflow_ctx *ctx, const unsigned char *src, size_t src_ps, int n_frames, int src_rows,
                                 int src_cols, size_t src_step_bytes, unsigned char *dst, size_t dst_ps, int dst_rows,
                                 int dst_cols, size_t dst_step_bytes, void *stream)
{
    if (!ctx || !src || !dst || n_frames < 1 || src_rows < 1 || src_cols < 1 || dst_rows < 1 || dst_cols < 1) return LKFLOW_E_ARG;
    if (src_step_bytes < (size_t)src_cols * 3 || dst_step_bytes < (size_t)dst_cols * 3) return LKFLOW_E_ARG;
    if (n_frames > 65535 || dst_rows > 65535 || src_cols > (1 << 24) || src_rows > (1 << 24) || dst_cols > (1 << 24)) return LKFLOW_E_ARG;
    int rc = set_device(ctx);
    if (rc) return rc;
    lk::ResizeArgs a{};
    a.src = src; a.src_plane = (long long)src_ps; a.src_pitch = (long long)src_step_bytes;
    a.dst = dst; a.dst_plane = (long long)dst_ps; a.dst_pitch = (long long)dst_step_bytes;
    a.srows = src_rows; a.scols = src_cols; a.drows = dst_rows; a.dcols = dst_cols;
    a.scale_x = 1.0 / ((double)dst_cols / src_cols);         // cv::resize: inv_scale = (double)dsize / ssize; scale = 1 / inv_scale
    a.scale_y = 1.0 / ((double)dst_rows / src_rows);
    hipLaunchKernelGGL(lk::resize_cubic_kernel, dim3((dst_cols + 255) / 256, dst_rows, n_frames), dim3(256), 0,
                       (hipStream_t)stream, a);
    HIP_TRY(ctx, hipGetLastError());
    return LKFLOW_OK;
}

int lkflow_frame_diff_u8_dev(lkflow_ctx *ctx, const unsigned char *cur, const unsigned char *prev, size_t src_ps,
                             int n_planes, int rows, int cols, size_t src_step_bytes, double rate,
                             unsigned char *diff, size_t dst_ps, size_t dst_step_bytes, void *stream)
{
    if (!ctx || !cur || !prev || !diff || rows < 1 || cols < 1 || n_planes < 1 || n_planes > 65535 || rows > 65535)
        return LKFLOW_E_ARG;
    if (src_step_bytes < (size_t)cols || dst_step_bytes < (size_t)cols) return LKFLOW_E_ARG;
    int rc = set_device(ctx);
    if (rc) return rc;
    lk::FrameDiffArgs a{};
    a.cur = cur; a.prev = prev; a.src_plane = (long long)src_ps; a.src_pitch = (long long)src_step_bytes;
    a.dst = diff; a.dst_plane = (long long)dst_ps; a.dst_pitch = (long long)dst_step_bytes;
    a.H = rows; a.W = cols; a.beta = (float)(-rate);
    hipLaunchKernelGGL(lk::frame_diff_kernel, dim3((cols + 255) / 256, rows, n_planes), dim3(256), 0,
                       (hipStream_t)stream, a);
    HIP_TRY(ctx, hipGetLastError());
    return LKFLOW_OK;
}

int lkflow_flow_centroid_f64_dev(lkflow_ctx *ctx, const double *U, const double *V, size_t uv_ps, int n_pairs,
                                 int rows, int cols, size_t uv_step_bytes, double thresh, double *out,
                                 void *stream)
{
    if (!ctx || !U || !V || !out || rows < 1 || cols < 1 || n_pairs < 1 || n_pairs > 65535) return LKFLOW_E_ARG;
    if (uv_step_bytes < (size_t)cols * 8 || uv_step_bytes % 8 || uv_ps % 8) return LKFLOW_E_ARG;
    int rc = set_device(ctx);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(ctx, hipMemsetAsync(out, 0, (size_t)n_pairs * 3 * sizeof(double), s));
    const long long ps = uv_ps / 8, st = uv_step_bytes / 8;
    // smallest double whose (correctly rounded) square root is >= thresh: sqrt(s) < thresh <=> s < thresh2.
    // thresh <= 0 or NaN: the reference's `val < thresh` is never true for val >= 0 -> every pixel counts.
    double thresh2;
    if (!(thresh > 0.0)) thresh2 = -INFINITY;
    else if (std::isinf(thresh)) thresh2 = INFINITY;
    else {
        thresh2 = thresh * thresh;
        if (std::isinf(thresh2)) thresh2 = DBL_MAX;
        while (thresh2 > 0.0 && std::sqrt(thresh2) >= thresh) thresh2 = std::nextafter(thresh2, 0.0);
        while (std::sqrt(thresh2) < thresh) thresh2 = std::nextafter(thresh2, INFINITY);
    }
    int gx = (cols + 255) / 256;
    int gy = rows < 32 ? rows : 32;
    hipLaunchKernelGGL(lk::centroid_kernel, dim3(gx, gy, n_pairs), dim3(256), 0, s, CPlanes{U, ps, st},
                       CPlanes{V, ps, st}, rows, cols, thresh2, out);
    HIP_TRY(ctx, hipGetLastError());
    return LKFLOW_OK;
}

}  // extern "C"

#include "lk_stream.h"
