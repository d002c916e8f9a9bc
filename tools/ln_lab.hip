// Development harness: times ro_layernorm_bwd on the training-step shape (not part of the library).
#include <cstdarg>
#include <vector>
#ifndef LAB_SRC
#define LAB_SRC "../few-shot-transformer-tts_amd/csrc/rowops.hip"
#endif
#include LAB_SRC
thread_local char g_b2s_err[512] = "";
int b2s_fail(const char* file, int line, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_b2s_err, sizeof(g_b2s_err), fmt, ap); va_end(ap);
    fprintf(stderr, "FAIL %s:%d: %s\n", file, line, g_b2s_err); return 1;
}
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 8148, D = argc > 2 ? atoi(argv[2]) : 768, iters = 200;
    void* dy; float *x, *gamma, *mean, *rstd, *dx, *dg, *db, *ws;
    hipMalloc(&dy, (size_t)M * D * 4); hipMalloc(&x, (size_t)M * D * 4); hipMalloc(&dx, (size_t)M * D * 4);
    hipMalloc(&gamma, D * 4); hipMalloc(&mean, M * 4); hipMalloc(&rstd, M * 4); hipMalloc(&dg, D * 4); hipMalloc(&db, D * 4);
    hipMalloc(&ws, (size_t)RO_LN_WS_ROWS * 2 * D * 4);
    hipMemset(dy, 0, (size_t)M * D * 4); hipMemset(x, 0, (size_t)M * D * 4); hipMemset(dx, 0, (size_t)M * D * 4);
    hipMemset(gamma, 0, D * 4); hipMemset(mean, 0, M * 4); hipMemset(rstd, 0, M * 4); hipMemset(dg, 0, D * 4); hipMemset(db, 0, D * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        for (int w = 0; w < 3; ++w) ro_layernorm_bwd(1, dy, 0, D, x, gamma, mean, rstd, dx, mode != 1, dg, db, M, D, nullptr, 1, 0, mode == 2 ? nullptr : ws);
        hipEventRecord(e0, 0);
        for (int i = 0; i < iters; ++i) ro_layernorm_bwd(1, dy, 0, D, x, gamma, mean, rstd, dx, mode != 1, dg, db, M, D, nullptr, 1, 0, mode == 2 ? nullptr : ws);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)M * D * (2 + 4 + 4 + (mode != 1 ? 4 : 0));
        printf("ln_bwd mode %d (%s): %.2f us/launch pair, %.2f TB/s\n", mode, mode == 0 ? "accumulate, ws" : mode == 1 ? "no accumulate, ws" : "accumulate, atomics",
               ms * 1e3 / iters, bytes / (ms * 1e-3 / iters) / 1e12);
    }
    return 0;
}
