// Development harness for csrc/attention32.hip: the 32x32x16 attention kernels against the 16-row kernels of attention.hip on the training-step
// shapes -- results (same inputs, same dropout masks) and time per launch.  Not part of the library.
//   build:  tools/build_attn32_lab.sh            (links attention.o + attention32.o built with -DB2S_LAB)
//   run:    B2S_LAB_ATTN32=0 tools/bin/attn32_lab   (b2s_flash_fwd / _bwd then reach the OLD kernels; the new ones are called directly)
#include <cstdarg>
#include <cmath>
#include <vector>
#include "../few-shot-transformer-tts_amd/csrc/b2s_common.h"
#include "../few-shot-transformer-tts_amd/csrc/attention.h"
thread_local char g_b2s_err[512] = "";
int b2s_fail(const char* file, int line, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_b2s_err, sizeof(g_b2s_err), fmt, ap); va_end(ap);
    fprintf(stderr, "FAIL %s:%d: %s\n", file, line, g_b2s_err); return 1;
}
struct Case { const char* name; int B, H, Lq, Lk, dh, mask; bool cross; bool skip; };
static double maxdiff(const std::vector<bf16_t>& x, const std::vector<bf16_t>& y, size_t rows, int ld, int col0, int ncol, double* ref) {
    double m = 0, r = 0;
    for (size_t i = 0; i < rows; ++i) for (int c = 0; c < ncol; ++c) {
        const double a = bf2f(x[i * ld + col0 + c]), b = bf2f(y[i * ld + col0 + c]);
        if (std::isnan(a) || std::isnan(b)) return 1e30;
        m = fmax(m, fabs(a - b)); r = fmax(r, fabs(a));
    }
    *ref = r; return m;
}
int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    const Case cases[] = {{"dec self  (causal)", 14, 8, 582, 582, 96, 2, false, false}, {"dec cross (klen)", 14, 8, 582, 114, 96, 1, true, false},
                          {"dec self  (causal,qskip)", 14, 8, 582, 582, 96, 2, false, true}, {"dec cross (klen,qskip)", 14, 8, 582, 114, 96, 1, true, true},
                          {"c3 cross (klen 256)", 14, 8, 582, 256, 96, 1, true, false}, {"enc self  (klen, dh 64)", 14, 8, 114, 114, 64, 1, false, false},
                          {"tiny (dh 32, Lq 37, Lk 53)", 3, 4, 37, 53, 32, 1, true, false}, {"tiny causal (dh 32, 70)", 3, 4, 70, 70, 32, 3, false, false}};
    const size_t n = (size_t)14 * 582 * 3 * 768;
    std::vector<bf16_t> h(n), hd(n / 3);
    uint32_t s = 1; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = f2bf((((s >> 8) & 0xffff) / 65536.f - 0.5f) * 2.0f); }
    for (auto& v : hd) { s = s * 1664525u + 1013904223u; v = f2bf((((s >> 8) & 0xffff) / 65536.f - 0.5f) * 0.5f); }
    bf16_t *qkv, *dqkv[2], *ctx[2], *dctx; float *lse[2], *dsum; int *klen, *qskip;
    hipMalloc(&qkv, n * 2); hipMalloc(&dctx, n * 2);
    for (int i = 0; i < 2; ++i) { hipMalloc(&dqkv[i], n * 2); hipMalloc(&ctx[i], n * 2); hipMalloc(&lse[i], 14 * 8 * 582 * 4); }
    hipMalloc(&dsum, 14 * 8 * 582 * 4); hipMalloc(&klen, 64); hipMalloc(&qskip, 64);
    hipMemcpy(qkv, h.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(dctx, hd.data(), n * 2 / 3, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int ci = -1;
    for (const Case& c : cases) for (int drop = 0; drop < 2; ++drop) {
        ++ci;
        if (only >= 0 && ci / 2 != only) continue;
        const int D = c.H * c.dh;
        int hl[14], hq[14];
        for (int b = 0; b < 14; ++b) { hl[b] = c.Lk - (b * (c.Lk / 5)) / 14; hq[b] = c.Lq - (b * (c.Lq / 5)) / 14; }
        hipMemcpy(klen, hl, sizeof(hl), hipMemcpyHostToDevice); hipMemcpy(qskip, hq, sizeof(hq), hipMemcpyHostToDevice);
        const size_t rowsq = (size_t)c.B * c.Lq, rowsk = (size_t)c.B * c.Lk;
        AttnArgs a;
        if (c.cross) { a.q = qkv; a.ldq = D; a.k = qkv + (size_t)14 * 582 * D; a.ldk = 2 * D; a.v = (const bf16_t*)a.k + D; a.ldv = 2 * D; }
        else { a.q = qkv; a.k = qkv + D; a.v = qkv + 2 * D; a.ldq = a.ldk = a.ldv = 3 * D; }
        a.B = c.B; a.H = c.H; a.Lq = c.Lq; a.Lk = c.Lk; a.scale = 1.f / sqrtf((float)c.dh); a.mask_mode = c.mask; a.klen = klen;
        a.drop = make_drop(drop ? 0.1f : 0.f, 1234, 7); a.ldo = D; a.dout = dctx; a.dsum = dsum;
        if (c.skip) a.qskip = qskip;
        std::vector<bf16_t> o[2], g[2]; std::vector<float> L[2];
        double t_us[2][3] = {{0}};
        for (int v = 0; v < 2; ++v) {                                 // v = 0: the 16-row kernels, 1: the 32x32 kernels
            AttnArgs x = a;
            x.lse = lse[v]; x.out = ctx[v];
            if (c.cross) { x.dq = dqkv[v]; x.lddq = D; x.dk = dqkv[v] + (size_t)14 * 582 * D; x.lddk = 2 * D; x.dv = (bf16_t*)x.dk + D; x.lddv = 2 * D; }
            else { x.dq = dqkv[v]; x.dk = dqkv[v] + D; x.dv = dqkv[v] + 2 * D; x.lddq = x.lddk = x.lddv = 3 * D; }
            hipMemset(ctx[v], 0x7f, n * 2); hipMemset(dqkv[v], 0x7f, n * 2);
            auto fwd = [&] { return v ? b2s_flash32_launch(x, c.dh, 0, 0) : b2s_flash_fwd(1, x, c.dh, 0); };
            AttnArgs xb = x; xb.oref = ctx[v];
            auto bdq = [&] { return v ? b2s_flash32_launch(xb, c.dh, 1, 0) : b2s_flash_bwd(1, x, c.dh, ctx[v], 0); };
            auto bkv = [&] { return v ? b2s_flash32_launch(xb, c.dh, 2, 0) : 0; };
            if (fwd() || bdq() || bkv()) return 1;
            if (hipDeviceSynchronize() != hipSuccess) { printf("kernel fault: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
            o[v].resize(n); g[v].resize(n); L[v].resize((size_t)c.B * c.H * c.Lq);
            hipMemcpy(o[v].data(), ctx[v], n * 2, hipMemcpyDeviceToHost); hipMemcpy(g[v].data(), dqkv[v], n * 2, hipMemcpyDeviceToHost);
            hipMemcpy(L[v].data(), lse[v], L[v].size() * 4, hipMemcpyDeviceToHost);
            const int it = 50; float ms;
            for (int w = 0; w < 3; ++w) fwd();
            hipEventRecord(e0, 0); for (int i = 0; i < it; ++i) fwd(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); t_us[v][0] = ms * 1e3 / it;
            for (int w = 0; w < 3; ++w) bdq();
            hipEventRecord(e0, 0); for (int i = 0; i < it; ++i) bdq(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); t_us[v][1] = ms * 1e3 / it;
            if (v) {
                for (int w = 0; w < 3; ++w) bkv();
                hipEventRecord(e0, 0); for (int i = 0; i < it; ++i) bkv(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1); t_us[v][2] = ms * 1e3 / it;
            }
        }
        double r0, r1, r2, r3, dl = 0;
        // rows the old kernel leaves as padding (qskip) are zero in both; lse compared on all rows
        const double d_o = maxdiff(o[0], o[1], rowsq, D, 0, D, &r0);
        double d_q, d_k, d_v;
        if (c.cross) {
            d_q = maxdiff(g[0], g[1], rowsq, D, 0, D, &r1);
            std::vector<bf16_t> k0(g[0].begin() + (size_t)14 * 582 * D, g[0].end()), k1(g[1].begin() + (size_t)14 * 582 * D, g[1].end());
            d_k = maxdiff(k0, k1, rowsk, 2 * D, 0, D, &r2); d_v = maxdiff(k0, k1, rowsk, 2 * D, D, D, &r3);
        } else {
            d_q = maxdiff(g[0], g[1], rowsq, 3 * D, 0, D, &r1); d_k = maxdiff(g[0], g[1], rowsk, 3 * D, D, D, &r2); d_v = maxdiff(g[0], g[1], rowsk, 3 * D, 2 * D, D, &r3);
        }
        for (size_t i = 0; i < L[0].size(); ++i) dl = fmax(dl, fabs((double)L[0][i] - L[1][i]));
        const double fl = 4.0 * c.B * c.H * (double)c.Lq * c.Lk * c.dh * ((c.mask & 2) ? 0.5 : 1.0);
        printf("%-26s drop=%d | diff out %.3g/%.2g lse %.2g dq %.3g/%.2g dk %.3g/%.2g dv %.3g/%.2g | old fwd %6.1f bwd %6.1f us | new fwd %6.1f (%5.0f TF) dq %6.1f (%5.0f) dkv %6.1f (%5.0f) us\n",
               c.name, drop, d_o, r0, dl, d_q, r1, d_k, r2, d_v, r3, t_us[0][0], t_us[0][1], t_us[1][0], fl / t_us[1][0] / 1e6, t_us[1][1],
               1.5 * fl / t_us[1][1] / 1e6, t_us[1][2], 2.0 * fl / t_us[1][2] / 1e6);
        fflush(stdout);
    }
    return 0;
}
