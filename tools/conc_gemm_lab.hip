// Do two GEMM chains on two HIP streams overlap?  (development tool)  Each chain = `n` back-to-back launches of an M x N x K GEMM
// that fills only part of the chip; one chain alone vs two chains at the same time.
#include <cstdarg>
#include <cstdlib>
#include <vector>
#include "../few-shot-transformer-tts_amd/csrc/gemm_glds.hip"
#include "../few-shot-transformer-tts_amd/csrc/gemm_glds256.hip"
thread_local char g_b2s_err[512] = "";
int b2s_fail(const char* file, int line, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, " (%s:%d)\n", file, line); return 1; }
int main(int argc, char** argv) {
    const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]), n = argc > 4 ? atoi(argv[4]) : 40;
    bf16_t *dA[2], *dB[2]; void* dC[2];
    hipStream_t st[2];
    for (int i = 0; i < 2; ++i) {
        hipMalloc(&dA[i], (size_t)M * K * 2); hipMalloc(&dB[i], (size_t)N * K * 2); hipMalloc(&dC[i], (size_t)M * N * 2);
        hipMemset(dA[i], 0x11, (size_t)M * K * 2); hipMemset(dB[i], 0x11, (size_t)N * K * 2);
        hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
    }
    auto launch = [&](int i) {
        GemmArgs g; g.M = M; g.N = N; g.K = K; g.A.p = dA[i]; g.A.ld = K; g.A.R = M; g.A.C = K; g.B.p = dB[i]; g.B.ld = K; g.B.R = N; g.B.C = K;
        g.C = dC[i]; g.ldc = N; g.c_fp32 = 0;
        return b2s_gemm_glds_launch(g, false, false, st[i]);
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 5; ++w) { launch(0); launch(1); }
    hipDeviceSynchronize();
    for (int mode = 0; mode < 2; ++mode) {
        hipEventRecord(e0, st[0]);
        hipStreamWaitEvent(st[1], e0, 0);
        for (int k = 0; k < n; ++k) { launch(0); if (mode) launch(1); }
        if (mode) { hipEvent_t j; hipEventCreate(&j); hipEventRecord(j, st[1]); hipStreamWaitEvent(st[0], j, 0); }
        hipEventRecord(e1, st[0]); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("M=%d N=%d K=%d: %s: %.2f us per launch%s\n", M, N, K, mode ? "two streams" : "one stream ", ms * 1e3 / n, mode ? " pair" : "");
    }
    return 0;
}
