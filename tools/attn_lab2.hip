// Development harness (not part of the library): A/B of the fused attention kernel variants on the training-step shapes -- times every
// variant and checks each against the 64-row / 64-key baseline variant on the same inputs (outputs, lse, dq, dk, dv).
#include <cstdarg>
#include <vector>
#include <cmath>
#include "../few-shot-transformer-tts_amd/csrc/attention.hip"
thread_local char g_b2s_err[512] = "";
int b2s_fail(const char* file, int line, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_b2s_err, sizeof(g_b2s_err), fmt, ap); va_end(ap);
    fprintf(stderr, "FAIL %s:%d: %s\n", file, line, g_b2s_err); return 1;
}
struct Case { const char* name; int B, H, Lq, Lk, dh, mask; bool cross; };
template <int DH, int RB> void run_fwd(const AttnArgs& a) { hipLaunchKernelGGL((attn_fwd_kernel<bf16_t, DH, RB>), dim3(cdiv(a.Lq, 64 * RB), a.B * a.H), dim3(256), 0, 0, a); }
template <int DH, int RB> void run_dq(const AttnArgs& a) { hipLaunchKernelGGL((attn_bwd_dq_kernel<bf16_t, DH, RB>), dim3(cdiv(a.Lq, 64 * RB), a.B * a.H), dim3(256), 0, 0, a); }
template <int DH, int KB> void run_dkv(const AttnArgs& a) { hipLaunchKernelGGL((attn_bwd_dkv_kernel<bf16_t, DH, KB>), dim3(cdiv(a.Lk, 64 * KB), a.B * a.H), dim3(256), 0, 0, a); }
static double maxdiff_bf16(const bf16_t* a, const bf16_t* b, size_t n, double* ref) {
    std::vector<bf16_t> ha(n), hb(n);
    hipMemcpy(ha.data(), a, n * 2, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, n * 2, hipMemcpyDeviceToHost);
    double d = 0, r = 0;
    for (size_t i = 0; i < n; ++i) { const double x = bf2f(ha[i]), y = bf2f(hb[i]); if (std::isnan(x) || std::isnan(y)) return 1e30; d = std::max(d, std::fabs(x - y)); r = std::max(r, std::fabs(y)); }
    *ref = r; return d;
}
template <typename F> double time_us(F f, int it = 100) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) f();
    hipEventRecord(e0, 0); for (int i = 0; i < it; ++i) f(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1e3 / it;
}
template <int DH> void run_case(const Case& c, bf16_t* qkv, bf16_t* dqkv, bf16_t* dqkv2, bf16_t* ctx, bf16_t* ctx2, bf16_t* dctx, float* lse, float* dsum, int* klen, int drop) {
    const int D = c.H * DH;
    const size_t nrow = (size_t)14 * 582;
    AttnArgs a;
    if (c.cross) { a.q = qkv; a.ldq = D; a.k = qkv + nrow * D; a.ldk = 2 * D; a.v = (const bf16_t*)a.k + D; a.ldv = 2 * D; }
    else { a.q = qkv; a.k = qkv + D; a.v = qkv + 2 * D; a.ldq = a.ldk = a.ldv = 3 * D; }
    a.B = c.B; a.H = c.H; a.Lq = c.Lq; a.Lk = c.Lk; a.scale = 1.f / sqrtf((float)DH); a.mask_mode = c.mask; a.klen = klen;
    a.drop = make_drop(drop ? 0.1f : 0.f, 1234, 7); a.lse = lse; a.out = ctx; a.ldo = D; a.oref = ctx;
    a.dout = dctx; a.dsum = dsum;
    auto set_d = [&](AttnArgs& x, bf16_t* base) {
        if (c.cross) { x.dq = base; x.lddq = D; x.dk = base + nrow * D; x.lddk = 2 * D; x.dv = (bf16_t*)x.dk + D; x.lddv = 2 * D; }
        else { x.dq = base; x.dk = base + D; x.dv = base + 2 * D; x.lddq = x.lddk = x.lddv = 3 * D; }
    };
    set_d(a, dqkv);
    AttnArgs a2 = a; a2.out = ctx2; set_d(a2, dqkv2);
    const double fl = 4.0 * c.B * c.H * (double)c.Lq * c.Lk * DH * (c.mask == 2 ? 0.5 : 1.0);
    hipMemset(ctx, 0, nrow * 768 * 2); hipMemset(ctx2, 0, nrow * 768 * 2); hipMemset(dqkv, 0, nrow * 3 * 768 * 2); hipMemset(dqkv2, 0, nrow * 3 * 768 * 2);
    run_fwd<DH, 1>(a); run_fwd<DH, 2>(a2); hipDeviceSynchronize();
    double r0, r1, r2, r3;
    const double d_out = maxdiff_bf16(ctx2, ctx, nrow * D, &r0);
    run_dq<DH, 1>(a); run_dkv<DH, 1>(a);
    a2.out = ctx; a2.oref = ctx;                    // same forward result for both backward variants
    run_dq<DH, 2>(a2); run_dkv<DH, 2>(a2); hipDeviceSynchronize();
    const size_t nall = c.cross ? nrow * 3 * D : nrow * 3 * D;
    const double d_all = maxdiff_bf16(dqkv2, dqkv, nall, &r1);
    (void)r2; (void)r3;
    const double f1 = time_us([&] { run_fwd<DH, 1>(a); }), f2 = time_us([&] { run_fwd<DH, 2>(a); });
    const double q1 = time_us([&] { run_dq<DH, 1>(a); }), q2 = time_us([&] { run_dq<DH, 2>(a); });
    const double k1 = time_us([&] { run_dkv<DH, 1>(a); }), k2 = time_us([&] { run_dkv<DH, 2>(a); });
    printf("%-20s drop=%d  fwd RB1 %6.2f RB2 %6.2f us | dq RB1 %6.2f RB2 %6.2f | dkv KB1 %6.2f KB2 %6.2f | (fwd %.0f TF best)  max|d out| %.3e (ref %.2f)  max|d dqkv| %.3e (ref %.2f)\n",
           c.name, drop, f1, f2, q1, q2, k1, k2, fl / std::min(f1, f2) / 1e6, d_out, r0, d_all, r1);
}
int main(int argc, char** argv) {
    const bool prof = argc > 1;          // profiling run (rocprofv3 --pmc): the decoder self-attention kernels only, a few launches each
    const Case cases[] = {{"dec self  (causal)", 14, 8, 582, 582, 96, 2, false}, {"dec cross (klen)", 14, 8, 582, 114, 96, 1, true},
                          {"enc self  (klen)", 14, 8, 114, 114, 64, 1, false}};
    const size_t n = (size_t)14 * 582 * 3 * 768;
    std::vector<bf16_t> h(n);
    uint32_t s = 1; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = f2bf((((s >> 8) & 0xffff) / 65536.f - 0.5f) * 2.0f); }
    bf16_t *qkv, *dqkv, *dqkv2, *ctx, *ctx2, *dctx; float *lse, *dsum; int* klen;
    hipMalloc(&qkv, n * 2); hipMalloc(&dqkv, n * 2); hipMalloc(&dqkv2, n * 2); hipMalloc(&ctx, n * 2); hipMalloc(&ctx2, n * 2); hipMalloc(&dctx, n * 2);
    hipMalloc(&lse, 14 * 8 * 582 * 4); hipMalloc(&dsum, 14 * 8 * 582 * 4); hipMalloc(&klen, 64);
    hipMemcpy(qkv, h.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(dctx, h.data() + 12345, n * 2 / 3, hipMemcpyHostToDevice);
    int hl[14]; for (int b = 0; b < 14; ++b) hl[b] = 114 - (b * 23) / 14; hipMemcpy(klen, hl, sizeof(hl), hipMemcpyHostToDevice);
    if (prof) {
        const Case& c = cases[0];
        const int D = c.H * 96;
        AttnArgs a;
        a.q = qkv; a.k = qkv + D; a.v = qkv + 2 * D; a.ldq = a.ldk = a.ldv = 3 * D;
        a.B = c.B; a.H = c.H; a.Lq = c.Lq; a.Lk = c.Lk; a.scale = 1.f / sqrtf(96.f); a.mask_mode = c.mask; a.klen = klen;
        a.lse = lse; a.out = ctx; a.ldo = D; a.oref = ctx; a.dout = dctx; a.dsum = dsum;
        a.dq = dqkv; a.dk = dqkv + D; a.dv = dqkv + 2 * D; a.lddq = a.lddk = a.lddv = 3 * D;
        for (int drop = 0; drop < 2; ++drop) {
            a.drop = make_drop(drop ? 0.1f : 0.f, 1234, 7);
            for (int i = 0; i < 3; ++i) { run_fwd<96, 1>(a); run_dq<96, 1>(a); run_dkv<96, 1>(a); run_fwd<96, 2>(a); }
        }
        hipDeviceSynchronize();
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep)
    for (const Case& c : cases) for (int drop = 0; drop < 2; ++drop) {
        if (c.dh == 96) run_case<96>(c, qkv, dqkv, dqkv2, ctx, ctx2, dctx, lse, dsum, klen, drop);
        else run_case<64>(c, qkv, dqkv, dqkv2, ctx, ctx2, dctx, lse, dsum, klen, drop);
    }
    return 0;
}
