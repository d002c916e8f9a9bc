#!/usr/bin/env python3
"""The persistent GEMM kernel (csrc/gemm_glds256.hip) draws its tickets with an atomic whose result register is read two K steps later (inline asm: the
compiler does not know the register is written asynchronously).  That is only correct while the compiler keeps the value in ONE register from the atomic
to the read and writes nothing else to it in between.  This script compiles the file to ISA and checks exactly that for every instantiation:
    python tools/check_persist_isa.py        (about a minute; exit code 1 on a violation)
Run it after any change to the producer part of gemm_glds256_persist_kernel or a compiler update."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "few-shot-transformer-tts_amd", "csrc", "gemm_glds256.hip")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "g.s")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-S", "--cuda-device-only", src, "-o", out],
                          cwd=os.path.dirname(src), stderr=subprocess.DEVNULL)
    s = open(out).read()
bad = 0
for m in re.finditer(r'^(_ZN4t25627gemm_glds256_persist_kernel\w+):', s, re.M):
    body = s[m.start():s.index('.Lfunc_end', m.start())].split('\n')
    draws = [(n, l.strip()) for n, l in enumerate(body) if re.match(r'\s*global_atomic_add v\d+, v\[', l)]
    regs = set(re.match(r'global_atomic_add (v\d+)', t).group(1) for _, t in draws)
    lo, hi = draws[0][0], draws[-1][0]
    writers = [t.strip() for n, t in enumerate(body) if lo < n < hi and re.match(r'\s*(v_|ds_read|global_load|scratch_load|flat_load)\w* (%s),' % '|'.join(regs), t)
               and not re.match(r'\s*v_mov_b32_e32 v\d+, 1$', t)]
    copies = [t.strip() for n, t in enumerate(body) if lo < n < hi + 200 and re.match(r'\s*v_mov_b32_e32 v\d+, (%s)$' % '|'.join(regs), t)]
    reads = [n for n, t in enumerate(body) if lo < n < hi + 200 and re.search(r'v_readfirstlane_b32 s\d+, (%s)$' % '|'.join(regs), t.strip())]
    flat = [t.strip() for t in body if 'flat_' in t or 'scratch_' in t]
    ok = len(regs) == 1 and not writers and not copies and reads and not flat
    print("%-60s register %s  draws %d  landing reads %d  other writers %d  copies %d  flat/scratch %d  %s" % (
        m.group(1)[-58:], sorted(regs), len(draws), len(reads), len(writers), len(copies), len(flat), "ok" if ok else "VIOLATION"))
    bad += 0 if ok else 1
sys.exit(1 if bad else 0)
