// Development harness: times the fused attention kernels on the training-step shapes (not part of the library).
#include <cstdarg>
#include <vector>
#ifndef LAB_SRC
#define LAB_SRC "../few-shot-transformer-tts_amd/csrc/attention.hip"
#endif
#include LAB_SRC
thread_local char g_b2s_err[512] = "";
int b2s_fail(const char* file, int line, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_b2s_err, sizeof(g_b2s_err), fmt, ap); va_end(ap);
    fprintf(stderr, "FAIL %s:%d: %s\n", file, line, g_b2s_err); return 1;
}
struct Case { const char* name; int B, H, Lq, Lk, dh, mask; bool cross; };
int main() {
    const Case cases[] = {{"dec self  (causal)", 14, 8, 582, 582, 96, 2, false}, {"dec cross (klen)", 14, 8, 582, 114, 96, 1, true},
                          {"enc self  (klen)", 14, 8, 114, 114, 64, 1, false}};
    const size_t n = (size_t)14 * 582 * 3 * 768;
    std::vector<bf16_t> h(n);
    uint32_t s = 1; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = f2bf((((s >> 8) & 0xffff) / 65536.f - 0.5f) * 0.5f); }
    bf16_t *qkv, *dqkv, *ctx, *dctx; float *lse, *dsum; int* klen;
    hipMalloc(&qkv, n * 2); hipMalloc(&dqkv, n * 2); hipMalloc(&ctx, n * 2); hipMalloc(&dctx, n * 2);
    hipMalloc(&lse, 14 * 8 * 582 * 4); hipMalloc(&dsum, 14 * 8 * 582 * 4); hipMalloc(&klen, 64);
    hipMemcpy(qkv, h.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(dctx, h.data(), n * 2 / 3, hipMemcpyHostToDevice);
    int hl[14]; for (int b = 0; b < 14; ++b) hl[b] = 114 - (b * 23) / 14; hipMemcpy(klen, hl, sizeof(hl), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (const Case& c : cases) for (int drop = 0; drop < 2; ++drop) {
        const int D = c.H * c.dh;
        AttnArgs a;
        if (c.cross) { a.q = qkv; a.ldq = D; a.k = qkv + (size_t)14 * 582 * D; a.ldk = 2 * D; a.v = (const bf16_t*)a.k + D; a.ldv = 2 * D; }
        else { a.q = qkv; a.k = qkv + D; a.v = qkv + 2 * D; a.ldq = a.ldk = a.ldv = 3 * D; }
        a.B = c.B; a.H = c.H; a.Lq = c.Lq; a.Lk = c.Lk; a.scale = 1.f / sqrtf((float)c.dh); a.mask_mode = c.mask; a.klen = klen;
        a.drop = make_drop(drop ? 0.1f : 0.f, 1234, 7); a.lse = lse; a.out = ctx; a.ldo = D;
        a.dout = dctx; a.dsum = dsum;
        if (c.cross) { a.dq = dqkv; a.lddq = D; a.dk = dqkv + (size_t)14 * 582 * D; a.lddk = 2 * D; a.dv = (bf16_t*)a.dk + D; a.lddv = 2 * D; }
        else { a.dq = dqkv; a.dk = dqkv + D; a.dv = dqkv + 2 * D; a.lddq = a.lddk = a.lddv = 3 * D; }
        double fl = 4.0 * c.B * c.H * (double)c.Lq * c.Lk * c.dh * (c.mask == 2 ? 0.5 : 1.0);
        float ms;
        const int it = 100;
        for (int w = 0; w < 3; ++w) b2s_flash_fwd(1, a, c.dh, 0);
        hipEventRecord(e0, 0); for (int i = 0; i < it; ++i) b2s_flash_fwd(1, a, c.dh, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double fwd_us = ms * 1e3 / it;
        for (int w = 0; w < 3; ++w) b2s_flash_bwd(1, a, c.dh, ctx, 0);
        hipEventRecord(e0, 0); for (int i = 0; i < it; ++i) b2s_flash_bwd(1, a, c.dh, ctx, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double bwd_us = ms * 1e3 / it;
        printf("%-20s drop=%d  fwd %7.2f us (%6.1f TF)   bwd(dq+dkv) %7.2f us (%6.1f TF)\n", c.name, drop, fwd_us, fl / fwd_us / 1e6, bwd_us,
               2.5 * fl / bwd_us / 1e6);
    }
    return 0;
}
