#!/bin/bash
# Development tool: a copy of libb2s_hip.so whose fused FFN decode kernel records shader-clock stamps at its phase boundaries
# (stamps of the last frame are printed at process exit).  The stamps are patched into a temporary copy of decode_fused.hip;
# the production source has none.  usage: bash tools/build_stamped_decode.sh ; then run with the lib swapped (see tools/gpu_dfstamp.sh)
set -e
cd "$(dirname "$0")/.."
C=few-shot-transformer-tts_amd/csrc
mkdir -p tools/bin
python - <<'PY'
src = open('few-shot-transformer-tts_amd/csrc/decode_fused.hip').read()
pre = '''
#include <cstdlib>
__device__ unsigned long long g_df_stamp[4096 * 8];
#define DF_STAMP(i) if (threadIdx.x == 0) g_df_stamp[(blockIdx.x & 4095) * 8 + (i)] = clock64();
static void df_dump() {
    static unsigned long long h[4096 * 8];
    (void)hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_df_stamp), sizeof(h)) != hipSuccess) return;
    double ph[7] = {0, 0, 0, 0, 0, 0, 0}; int n = 0;
    for (int w = 0; w < 256; ++w) { if (!h[w * 8 + 6]) continue; ++n; for (int p = 0; p < 6; ++p) ph[p] += (double)(h[w * 8 + p + 1] - h[w * 8 + p]); }
    if (n) fprintf(stderr, "k_df_ffn phases (cycles, mean of %d WGs): issue %.0f | x+LN %.0f | gemv1 %.0f | relu %.0f | gemv2 %.0f | store %.0f\\n", n, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n, ph[5] / n);
}
'''
src = src.replace('#include "decode_fused.h"\n', '#include "decode_fused.h"\n' + pre, 1)
a = src.index('    if constexpr (FAST && sizeof(T) == 2) {\n        // Default sizes, bf16: the slice')
body = src[a:]
def ins(body, anchor, stamp, before=False):
    assert anchor in body, anchor
    return body.replace(anchor, (stamp + anchor) if before else (anchor + stamp), 1)
body = ins(body, '        const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;\n', '        DF_STAMP(0)\n')
body = ins(body, '        __builtin_amdgcn_sched_barrier(0);\n        load_x_ln<T, UB, 8>', '        DF_STAMP(1)\n', before=True)
body = ins(body, '        load_x_ln<T, UB, 8>(c, b0, sl == 0, xs, hs, red, tid);\n', '        DF_STAMP(2)\n')
body = ins(body, '        __syncthreads();\n        const DropCfg dh_ = salted(a.drop_hid, t);\n', '        DF_STAMP(3)\n')
body = ins(body, '            TT<T>::st(fs + i, v);\n        }\n        __syncthreads();\n', '        DF_STAMP(4)\n')
body = ins(body, '        __syncthreads();\n        store_partial<UB>(c, sl, b0, xs, t, tid);\n', '        DF_STAMP(5)\n', before=False) if False else body
body = body.replace('        __syncthreads();\n        store_partial<UB>(c, sl, b0, xs, t, tid);\n    } else {', '        __syncthreads();\n        DF_STAMP(5)\n        store_partial<UB>(c, sl, b0, xs, t, tid);\n        DF_STAMP(6)\n    } else {', 1)
out = src[:a] + body
out = out.replace('int b2s_df_ffn(int dtype, const DfFfn& a, hipStream_t st) {', 'int b2s_df_ffn(int dtype, const DfFfn& a, hipStream_t st) {\n    static int calls = 0; if (++calls % 3000 == 0) df_dump();', 1)
open('tools/bin/decode_fused_stamped.hip', 'w').write(out)
PY
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -I$C -c tools/bin/decode_fused_stamped.hip -o tools/bin/decode_fused_stamped.o
hipcc --offload-arch=gfx950 -shared -fPIC $C/obj/gemm.o $C/obj/gemm_glds.o $C/obj/gemm_glds256.o $C/obj/gemm_skinny.o $C/obj/attention.o $C/obj/rowops.o $C/obj/engine.o $C/obj/capi_ops.o $C/obj/decode.o tools/bin/decode_fused_stamped.o -o tools/bin/libb2s_hip_stamped.so
echo built tools/bin/libb2s_hip_stamped.so
