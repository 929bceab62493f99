// Host-buffer entry points timed from C (no Python in the loop): lkflow_compute_f64 / lkflow_compute_u8 on one 1080p pair,
// malloc'ed (numpy-like, 16-byte aligned) vs page-aligned buffers, with and without LKFLOW_OPT_PIN_HOST.
//   g++ -O2 -I include -o tools/ubench/host_call tools/ubench/host_call.cpp -L optical-flow_amd -llkflow -Wl,-rpath,'$ORIGIN/../../optical-flow_amd'
#include "lkflow.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const int rows = 1080, cols = 1920, reps = argc > 1 ? atoi(argv[1]) : 10;
    const size_t npx = (size_t)rows * cols, nb = npx * 8;
    lkflow_ctx *ctx = nullptr;
    if (lkflow_ctx_create(&ctx, 0)) { printf("no context\n"); return 1; }
    // aligned: 0 = malloc-like (+16 bytes), 1 = page-aligned, 2 = MADV_NOHUGEPAGE (4 KiB pages), 3 = 2 MiB-aligned + MADV_HUGEPAGE
    for (int aligned = 0; aligned < 4; ++aligned) {
        char *raw[4], *p[4];
        for (int k = 0; k < 4; ++k) {
            const size_t cap = (nb + (4u << 20)) & ~((size_t)(2u << 20) - 1);
            raw[k] = (char *)aligned_alloc(2u << 20, cap);
            if (aligned >= 2) madvise(raw[k], cap, aligned == 2 ? MADV_NOHUGEPAGE : MADV_HUGEPAGE);
            p[k] = raw[k] + (aligned == 0 ? 16 : 0);
        }
        double *a = (double *)p[0], *b = (double *)p[1], *u = (double *)p[2], *v = (double *)p[3];
        for (size_t i = 0; i < npx; ++i) { a[i] = (double)((i * 2654435761u) >> 8 & 0xffff) / 65535.0; b[i] = a[(i + 1) % npx]; }
        memset(u, 0, nb); memset(v, 0, nb);
        unsigned char *a8 = (unsigned char *)malloc(npx), *b8 = (unsigned char *)malloc(npx);
        for (size_t i = 0; i < npx; ++i) { a8[i] = (unsigned char)(a[i] * 255); b8[i] = (unsigned char)(b[i] * 255); }
        for (int pin = 0; pin < 2; ++pin) {
            lkflow_ctx_set_option(ctx, LKFLOW_OPT_PIN_HOST, pin);
            for (int u8 = 0; u8 < 2; ++u8) {
                double tmin = 1e9, tsum = 0;
                for (int r = 0; r < reps + 1; ++r) {
                    const double t0 = now();
                    int rc = u8 ? lkflow_compute_u8(ctx, a8, b8, rows, cols, cols, u, v, (size_t)cols * 8, 3, 15)
                                : lkflow_compute_f64(ctx, a, b, rows, cols, (size_t)cols * 8, u, v, (size_t)cols * 8, 3, 15, 0);
                    const double t = now() - t0;
                    if (rc) { printf("rc=%d %s\n", rc, lkflow_last_error(ctx)); return 1; }
                    if (r) { tsum += t; if (t < tmin) tmin = t; }
                }
                printf("%s %s %s: mean %.3f ms  min %.3f ms\n", (aligned == 0 ? "malloc+16   " : aligned == 1 ? "page-aligned" : aligned == 2 ? "NOHUGEPAGE  " : "HUGEPAGE    "), pin ? "PIN_HOST" : "pageable", u8 ? "u8 " : "f64",
                       tsum / reps * 1e3, tmin * 1e3);
            }
        }
        lkflow_ctx_set_option(ctx, LKFLOW_OPT_PIN_HOST, 0);
        for (int k = 0; k < 4; ++k) free(raw[k]);
        free(a8); free(b8);
    }
    lkflow_ctx_destroy(ctx);
    return 0;
}
