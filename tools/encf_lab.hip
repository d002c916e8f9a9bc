// Phase timing of the fused encoder kernels (development tool): every workgroup keeps 100 MHz wall-clock stamps at its phase boundaries
// (ENCF_STAMPS in enc_fused.hip) and the host prints per-phase means, the dispatch ramp and the event time of the launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DENCF_STAMPS tools/encf_lab.hip -o tools/bin/encf_lab ; encf_lab [B] [S] [slab_bf16] [cold]
#include <cstdarg>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <hip/hip_runtime.h>
#include "../few-shot-transformer-tts_amd/csrc/enc_fused.hip"
thread_local char g_b2s_err[512] = "";
int b2s_fail(const char* file, int line, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, " (%s:%d)\n", file, line); return 1; }
__global__ void k_trash(float* p, long n) { for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) p[i] = p[i] * 1.0001f + 1.f; }
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 14, S = argc > 2 ? atoi(argv[2]) : 114, sb = argc > 3 ? atoi(argv[3]) : 0, cold = argc > 4 ? atoi(argv[4]) : 1;
    using namespace encf;
    const long M = (long)B * S;
    bf16_t *X, *Wqkv, *Wo, *W1, *W2, *qkv, *ctx, *f, *dz, *dqkv; float* lse; void* slabs; int* klen; float* trash; unsigned long long* dS;
    hipMalloc(&X, M * D * 2); hipMalloc(&Wqkv, 3L * D * D * 2); hipMalloc(&Wo, (long)D * D * 2); hipMalloc(&W1, (long)FF * D * 2); hipMalloc(&W2, (long)FF * D * 2);
    hipMalloc(&qkv, M * 3 * D * 2); hipMalloc(&ctx, M * D * 2); hipMalloc(&f, M * FF * 2); hipMalloc(&dz, M * FF * 2); hipMalloc(&dqkv, M * 3 * D * 2);
    hipMalloc(&lse, (long)B * NH * S * 4); hipMalloc(&slabs, 16 * M * D * 4); hipMalloc(&klen, B * 4); hipMalloc(&trash, 1L << 30); hipMalloc(&dS, 4096 * 64);
    hipMemset(X, 0x3c, M * D * 2); hipMemset(Wqkv, 0x3b, 3L * D * D * 2); hipMemset(Wo, 0x3b, (long)D * D * 2); hipMemset(W1, 0x3b, (long)FF * D * 2); hipMemset(W2, 0x3b, (long)FF * D * 2);
    hipMemset(qkv, 0x3c, M * 3 * D * 2); hipMemset(ctx, 0x3c, M * D * 2); hipMemset(f, 0x3c, M * FF * 2); hipMemset(lse, 0, (long)B * NH * S * 4); hipMemset(trash, 0, 1L << 30);
    std::vector<int> kl(B, S); hipMemcpy(klen, kl.data(), B * 4, hipMemcpyHostToDevice);
    hipMemcpyToSymbol(HIP_SYMBOL(g_encf_stamp), &dS, sizeof(dS));
    DropCfg dr = make_drop(0.1f, 1, 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[4] = {"attn fwd", "attn bwd", "ffn fwd", "ffn bwd"};
    const char* phases[4] = {"entry->stage0 | phase1 | core | Wo wait+barrier | out-proj | stores", "entry->stage0 | phase1 | role A | role B | phase3 | stores",
                             "entry->stage0 | phase1 | phase2 | stores", "entry->stage0 | phase1 | phase2 | stores"};
    for (int k = 0; k < 4; ++k) {
        const int nwg = B * 8, nph = k < 2 ? 6 : 4;
        double ph[8] = {0}, tot = 0, ev = 0, ramp = 0, span = 0; int n = 0;
        for (int it = 0; it < 14; ++it) {
            hipMemset(dS, 0, 4096 * 64);
            if (cold) hipLaunchKernelGGL(k_trash, dim3(2048), dim3(256), 0, 0, trash, (1L << 30) / 4);     // 1 GB through the L2s / the memory-side cache
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            if (k == 0) { EncfAttnFwd a{X, Wqkv, Wo, klen, B, S, dr, qkv, ctx, lse, slabs}; if (b2s_encf_attn_fwd(a, sb, 0)) return 1; }
            if (k == 1) { EncfAttnBwd a{X, qkv, ctx, lse, Wo, Wqkv, klen, B, S, dr, dqkv, slabs}; if (b2s_encf_attn_bwd(a, sb, 0)) return 1; }
            if (k == 2) { EncfFfn a{X, W1, W2, f, nullptr, slabs, B, S, dr, 1.f}; if (b2s_encf_ffn(a, false, sb, 0)) return 1; }
            if (k == 3) { EncfFfn a{X, W1, W2, f, dz, slabs, B, S, dr, 1.1f}; if (b2s_encf_ffn(a, true, sb, 0)) return 1; }
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it < 4) continue;
            std::vector<unsigned long long> h((size_t)nwg * 8);
            hipMemcpy(h.data(), dS, (size_t)nwg * 64, hipMemcpyDeviceToHost);
            unsigned long long w0 = ~0ull, w1 = 0, s1 = 0;
            for (int w = 0; w < nwg; ++w) { w0 = std::min(w0, h[w * 8]); s1 = std::max(s1, h[w * 8]); w1 = std::max(w1, h[w * 8 + nph]); }
            for (int w = 0; w < nwg; ++w) { for (int p = 0; p < nph; ++p) ph[p] += (double)(h[w * 8 + p + 1] - h[w * 8 + p]) * 0.01 / nwg; tot += (double)(h[w * 8 + nph] - h[w * 8]) * 0.01 / nwg; }
            ev += ms * 1e3; ramp += (s1 - w0) * 0.01; span += (w1 - w0) * 0.01; ++n;
        }
        printf("%-8s B=%d S=%d slab_bf16=%d %s: event %.1f us, first entry -> last exit %.1f us, dispatch ramp %.1f us, per-WG total %.1f us\n   phases (us) [%s]:", names[k], B, S, sb,
               cold ? "cold" : "warm", ev / n, span / n, ramp / n, tot / n, phases[k]);
        for (int p = 0; p < nph; ++p) printf(" %.2f", ph[p] / n);
        printf("\n");
    }
    return 0;
}
