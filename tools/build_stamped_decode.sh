#!/bin/bash
# Development tool: a copy of libb2s_hip.so whose fused decode kernels record wall-clock stamps (100 MHz) at their phase boundaries.
# The stamps are patched into a temporary copy of decode_fused.hip -- after every workgroup barrier at the top level of k_df_attn /
# k_df_ffn and after every helper call (finish_x_ln, gemv, attend, store_partial) -- so the production source carries none.
# usage: bash tools/build_stamped_decode.sh ; B2S_LIB_PATH=tools/bin/libb2s_hip_stamped.so python tools/dec_stamps.py
set -e
cd "$(dirname "$0")/.."
C=few-shot-transformer-tts_amd/csrc
mkdir -p tools/bin
python - <<'PY'
import re, json
src = open('few-shot-transformer-tts_amd/csrc/decode_fused.hip').read()
pre = '''
__device__ unsigned long long g_df_stamp[3][512][16];
#define DF_STAMP(kind) do { if (threadIdx.x == 0 && blockIdx.x < 512 && df_ph < 16) g_df_stamp[kind][blockIdx.x][df_ph] = wall_clock64(); ++df_ph; } while (0)
extern "C" int b2s_df_stamp_read(unsigned long long* host) {
    (void)hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_df_stamp), sizeof(unsigned long long) * 3 * 512 * 16);
}
'''
src = src.replace('#include "decode_fused.h"\n', '#include "decode_fused.h"\n' + pre, 1)
labels = {}
def patch(kname, kind_expr, key):
    global src
    a = src.index('void %s(' % kname)
    a = src.index('{\n', a) + 2
    # end of the kernel: the first line that is just "}" at column 0
    e = src.index('\n}\n', a)
    body = src[a:e]
    out, names = ['    int df_ph = 0; DF_STAMP(%s);\n' % kind_expr], ['start']
    lines = body.split('\n')
    i = 0
    while i < len(lines):
        ln = lines[i]
        out.append(ln + '\n')
        s = ln.strip()
        top = (len(ln) - len(ln.lstrip())) in (4, 8) and not ln.lstrip().startswith('//')
        hit = None
        if top and s.startswith('__syncthreads();'): hit = 'barrier'
        m = re.match(r'(finish_x_ln|load_x_ln|gemv|gemv_lds|attend|store_partial)<', s) if top else None
        if m:
            # statement may span lines: until parentheses balance and the line ends with ';'
            depth = ln.count('(') - ln.count(')')
            while depth > 0 or not lines[i].rstrip().endswith(';'):
                i += 1; out.append(lines[i] + '\n'); depth += lines[i].count('(') - lines[i].count(')')
            hit = m.group(1)
        if hit:
            ind = ' ' * (len(ln) - len(ln.lstrip()))
            out.append(ind + 'DF_STAMP(%s);\n' % kind_expr); names.append(hit)
        i += 1
    labels[key] = names
    src = src[:a] + ''.join(out).rstrip('\n') + src[e:]
patch('k_df_ffn', '2', 'ffn')
patch('k_df_attn', 'SELF ? 0 : 1', 'attn')
open('tools/bin/decode_fused_stamped.hip', 'w').write(src)
json.dump(labels, open('tools/bin/decode_fused_stamped.json', 'w'))
print(labels)
PY
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -I$C -c tools/bin/decode_fused_stamped.hip -o tools/bin/decode_fused_stamped.o
hipcc --offload-arch=gfx950 -shared -fPIC $C/obj/gemm.o $C/obj/gemm_glds.o $C/obj/gemm_glds256.o $C/obj/gemm_skinny.o $C/obj/attention.o $C/obj/rowops.o $C/obj/engine.o $C/obj/capi_ops.o $C/obj/decode.o tools/bin/decode_fused_stamped.o -o tools/bin/libb2s_hip_stamped.so
echo built tools/bin/libb2s_hip_stamped.so
