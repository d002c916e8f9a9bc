// Standalone bench / check harness for the bf16 GEMM kernels (development tool; not part of libb2s_hip.so).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DLAB_KERNEL='"path/to/kernel.hip"'] [-D...] tools/gemm_lab.hip -o gemm_lab
//   ./gemm_lab [iters]
// Times each training-step shape in NT / NN / TN form with hipEvents over `iters` back-to-back launches (hot clocks),
// and checks a sample of outputs against a double-precision host dot product.
#include <cstdarg>
#include <cstdlib>
#include <vector>
#include <string>
#ifndef LAB_KERNEL
#define LAB_KERNEL "../few-shot-transformer-tts_amd/csrc/gemm_glds.hip"
#endif
#include LAB_KERNEL
#ifndef LAB_NO256
#include "../few-shot-transformer-tts_amd/csrc/gemm_glds256.hip"
#endif

thread_local char g_b2s_err[512] = "";
int b2s_fail(const char* file, int line, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_b2s_err, sizeof(g_b2s_err), fmt, ap); va_end(ap);
    fprintf(stderr, "FAIL %s:%d: %s\n", file, line, g_b2s_err);
    return 1;
}
#ifndef LAB_LAUNCH
#define LAB_LAUNCH b2s_gemm_glds_launch
#endif

struct Shape { int M, N, K; int ta, tb; int splitk; const char* what; };

static uint32_t rng_state = 12345;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 65536.f - 0.5f; }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 50;
    const int only = argc > 2 ? atoi(argv[2]) : -1;
    std::vector<Shape> shapes = {
        {8148, 768, 768, 0, 0, 1, "fwd attn out / q proj"},
        {8148, 2304, 768, 0, 0, 1, "fwd qkv"},
        {8148, 3072, 768, 0, 0, 1, "fwd ffn in"},
        {8148, 768, 3072, 0, 0, 1, "fwd ffn out"},
        {1596, 2048, 512, 0, 0, 1, "fwd enc ffn in"},
        {1596, 1536, 512, 0, 0, 1, "fwd enc qkv"},
        {1596, 512, 512, 0, 0, 1, "fwd enc attn out"},
        {1596, 512, 1536, 0, 1, 1, "dX enc qkv"},
        {1596, 768, 1536, 0, 1, 1, "dX cross kv (old)"},
        {1596, 512, 2048, 0, 0, 1, "fwd enc ffn out"},
        {8148, 768, 768, 0, 1, 1, "dX attn out"},
        {8148, 768, 2304, 0, 1, 1, "dX qkv"},
        {8148, 768, 3072, 0, 1, 1, "dX ffn in"},
        {8148, 3072, 768, 0, 1, 1, "dX ffn out"},
        {768, 768, 8148, 1, 1, 10, "dW attn out"},
        {3072, 768, 8148, 1, 1, 3, "dW ffn in"},
        {768, 3072, 8148, 1, 1, 3, "dW ffn out"},
        {4096, 4096, 4096, 0, 0, 1, "4096^3 NT"},
        {8192, 8192, 8192, 0, 0, 1, "8192^3 NT"},
        {4096, 4096, 4096, 1, 1, 1, "4096^3 TT"},
        {4096, 4096, 4096, 0, 1, 1, "4096^3 NT(b T)"},
        {4096, 4096, 4096, 1, 0, 1, "4096^3 TN(a T)"},
    };
    if (const char* sp = getenv("LAB_SHAPES")) {          // "M,N,K,ta,tb;M,N,K,ta,tb;..." replaces the built-in list
        shapes.clear();
        std::string all(sp);
        size_t pos = 0;
        while (pos < all.size()) {
            size_t end = all.find(';', pos); if (end == std::string::npos) end = all.size();
            Shape sh{0, 0, 0, 0, 0, 1, "custom"};
            if (sscanf(all.substr(pos, end - pos).c_str(), "%d,%d,%d,%d,%d", &sh.M, &sh.N, &sh.K, &sh.ta, &sh.tb) == 5) shapes.push_back(sh);
            pos = end + 1;
        }
    }
    size_t maxel = (size_t)8192 * 8192;
    // LAB_ROT=n: n placements of every operand, used round-robin by the timed launches (n * footprint > 256 MB defeats the
    // memory-side cache, as inside the training step); LAB_EPI=1: fp32 output with bias + dropout + fp32 residual
    const int rot = getenv("LAB_ROT") ? atoi(getenv("LAB_ROT")) : 1;
    const bool epi = getenv("LAB_EPI") != nullptr;
    std::vector<bf16_t> hA(maxel), hB(maxel);
    for (size_t i = 0; i < maxel; ++i) { hA[i] = f2bf(frand()); hB[i] = f2bf(frand()); }
    bf16_t *dA, *dB; void* dC;
    float *dBias, *dRes;
    hipMalloc(&dA, maxel * 2 * rot); hipMalloc(&dB, maxel * 2 * rot); hipMalloc(&dC, maxel * 4 * rot);
    hipMalloc(&dBias, 8192 * 4); hipMalloc(&dRes, maxel * 4 * rot);
    hipMemset(dBias, 0, 8192 * 4); hipMemset(dRes, 0, maxel * 4 * rot);
    for (int r = 0; r < rot; ++r) {
        hipMemcpy(dA + r * maxel, hA.data(), maxel * 2, hipMemcpyHostToDevice);
        hipMemcpy(dB + r * maxel, hB.data(), maxel * 2, hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> hC;
    double tot_us = 0, tot_fl = 0;
    for (size_t si = 0; si < shapes.size(); ++si) {
        if (only >= 0 && (int)si != only) continue;
        const Shape& s = shapes[si];
        GemmArgs g;
        g.M = s.M; g.N = s.N; g.K = s.K;
        g.A.p = dA; g.B.p = dB;
        if (s.ta) { g.A.ld = s.M; g.A.R = s.K; g.A.C = s.M; } else { g.A.ld = s.K; g.A.R = s.M; g.A.C = s.K; }
        if (s.tb) { g.B.ld = s.N; g.B.R = s.K; g.B.C = s.N; } else { g.B.ld = s.K; g.B.R = s.N; g.B.C = s.K; }
        g.C = dC; g.ldc = s.N; g.splitk = s.splitk;
        if (s.splitk > 1) { g.c_fp32 = 1; g.epi.accumulate = 1; } else g.c_fp32 = 0;
        if (epi && s.splitk == 1) { g.c_fp32 = 1; g.epi.bias = dBias; g.epi.residual = dRes; g.epi.ldr = s.N; g.epi.drop = {0x1000000u, 1234u, 1.0f}; }
        auto place = [&](int it) {
            const int r = it % rot;
            g.A.p = dA + r * maxel; g.B.p = dB + r * maxel; g.C = (char*)dC + r * maxel * 4; if (g.epi.residual) g.epi.residual = dRes + r * maxel;
        };
        hipMemset(dC, 0, (size_t)s.M * s.N * 4);
        if (LAB_LAUNCH(g, s.ta, s.tb, 0)) return 1;
        hipDeviceSynchronize();
        // check a sample
        const size_t nC = (size_t)s.M * s.N;
        hC.resize(nC);
        if (g.c_fp32) hipMemcpy(hC.data(), dC, nC * 4, hipMemcpyDeviceToHost);
        else { std::vector<bf16_t> t(nC); hipMemcpy(t.data(), dC, nC * 2, hipMemcpyDeviceToHost); for (size_t i = 0; i < nC; ++i) hC[i] = bf2f(t[i]); }
        double worst = 0;
        for (int smp = 0; smp < 400; ++smp) {
            int m = (smp * 7919 + 13) % s.M, n = (smp * 104729 + 7) % s.N;
            if (smp < 8) { m = smp & 1 ? s.M - 1 - smp : smp; n = smp & 2 ? s.N - 1 - smp : smp; }
            double ref = 0;
            for (int k = 0; k < s.K; ++k) {
                float a = s.ta ? bf2f(hA[(size_t)k * s.M + m]) : bf2f(hA[(size_t)m * s.K + k]);
                float b = s.tb ? bf2f(hB[(size_t)k * s.N + n]) : bf2f(hB[(size_t)n * s.K + k]);
                ref += (double)a * b;
            }
            double err = fabs(ref - hC[(size_t)m * s.N + n]) / (0.02 * sqrt((double)s.K) * 0.083 + 0.01 * fabs(ref));
            if (err > worst) worst = err;
        }
        for (int w = 0; w < 5; ++w) LAB_LAUNCH(g, s.ta, s.tb, 0);
        hipEventRecord(e0, 0);
        for (int it = 0; it < iters; ++it) { place(it); LAB_LAUNCH(g, s.ta, s.tb, 0); }
        place(0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / iters, fl = 2.0 * s.M * s.N * s.K;
        printf("%2zu %-22s M=%5d N=%5d K=%5d %s%s sk=%2d  %8.2f us  %7.1f TF  err=%.2f%s\n", si, s.what, s.M, s.N, s.K, s.ta ? "T" : "N",
               s.tb ? "T" : "N", s.splitk, us, fl / us / 1e6, worst, worst > 1.0 ? "  <-- MISMATCH" : "");
        if (si < 17 || getenv("LAB_SHAPES")) { tot_us += us; tot_fl += fl; }
    }
    printf("step-shape mix: %.1f us total, %.1f TF/s\n", tot_us, tot_fl / tot_us / 1e6);
#ifndef LAB_NO256
    if (getenv("LAB_GROUPED")) {
        // the weight gradients of one decoder layer as the engine launches them: one grouped launch, 7 problems, full-depth K
        struct P { int M, N, K; };
        const P ps[7] = {{3072, 768, 8148}, {768, 3072, 8148}, {2304, 768, 8148}, {768, 768, 8148}, {768, 768, 8148}, {768, 768, 8148}, {1536, 768, 1596}};
        GemmArgs probs[7];
        size_t offA = 0, offB = 0, offC = 0;
        double fl = 0;
        for (int i = 0; i < 7; ++i) {
            GemmArgs& g = probs[i];
            g.M = ps[i].M; g.N = ps[i].N; g.K = ps[i].K;
            g.A.p = dA + offA; g.A.ld = g.M; g.A.R = g.K; g.A.C = g.M;
            g.B.p = dB + offB; g.B.ld = g.N; g.B.R = g.K; g.B.C = g.N;
            g.C = (float*)dC + offC; g.ldc = g.N; g.c_fp32 = 1; g.epi.accumulate = 1; g.splitk = 1;
            offA += (size_t)g.K * g.M; offB += (size_t)g.K * g.N; offC += (size_t)g.M * g.N;
            fl += 2.0 * g.M * g.N * g.K;
        }
        for (int w = 0; w < 3; ++w) b2s_gemm_glds256_grouped_launch(probs, 7, b2s_gemm_zero_page(), 0);
        hipEventRecord(e0, 0);
        for (int it = 0; it < iters; ++it) b2s_gemm_glds256_grouped_launch(probs, 7, b2s_gemm_zero_page(), 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("grouped decoder-layer dW (7 problems, 288 tiles): %.1f us  %.1f TF\n", ms * 1e3 / iters, fl / (ms * 1e3 / iters) / 1e6);
        // the same problems one by one with the engine's split-K choice
        const int sk[7] = {3, 3, 4, 14, 14, 14, 6};
        hipEventRecord(e0, 0);
        for (int it = 0; it < iters; ++it)
            for (int i = 0; i < 7; ++i) { GemmArgs g = probs[i]; g.splitk = sk[i]; LAB_LAUNCH(g, 1, 1, 0); }
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("same 7 problems, split-K launches + slab reduce: %.1f us  %.1f TF\n", ms * 1e3 / iters, fl / (ms * 1e3 / iters) / 1e6);
        // sensitivity: subsets of the group, and the same problems with a padded leading dimension (ld = 4096)
        auto time_group = [&](const char* what, GemmArgs* pp, int n) {
            double f = 0; for (int i = 0; i < n; ++i) f += 2.0 * pp[i].M * pp[i].N * pp[i].K;
            for (int w = 0; w < 2; ++w) b2s_gemm_glds256_grouped_launch(pp, n, b2s_gemm_zero_page(), 0);
            hipEventRecord(e0, 0);
            for (int it = 0; it < iters; ++it) b2s_gemm_glds256_grouped_launch(pp, n, b2s_gemm_zero_page(), 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1);
            printf("  %-44s %7.1f us  %6.1f TF\n", what, t * 1e3 / iters, f / (t * 1e3 / iters) / 1e6);
        };
        time_group("problems 0-1 (144 tiles)", probs, 2);
        time_group("problems 0-2 (198 tiles)", probs, 3);
        time_group("problems 0-5 (252 tiles, all long)", probs, 6);
        {
            GemmArgs padded[7]; size_t oa = 0, ob = 0;
            for (int i = 0; i < 7; ++i) {
                padded[i] = probs[i];
                padded[i].A.p = dA + oa; padded[i].A.ld = 4096; padded[i].B.p = dB + ob; padded[i].B.ld = 4096;
                oa += (size_t)padded[i].K * 4096; ob += (size_t)padded[i].K * 4096;
            }
            if (oa <= maxel * rot && ob <= maxel * rot) time_group("all 7, operands with ld = 4096", padded, 7);
            else printf("  (padded test needs LAB_ROT >= %zu)\n", oa / maxel + 1);
        }
        for (int i = 0; i < 3; ++i) {
            GemmArgs g = probs[i]; g.splitk = 1;
            hipEventRecord(e0, 0);
            for (int it = 0; it < iters; ++it) LAB_LAUNCH(g, 1, 1, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("  problem %d alone, sk=1 (%d tiles): %.1f us\n", i, (g.M / 256) * ((g.N + 127) / 128), ms * 1e3 / iters);
        }
    }
#endif
    return 0;
}
