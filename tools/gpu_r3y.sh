#!/bin/bash
out=$PWD/gpurun_out; : > $out/r3y_ab.txt
for r in 1 2; do
for arm in "B2S_X=0" "B2S_DW_STAGES=1" "B2S_DW_STAGES=3" "B2S_GEMM256_NB=4" "B2S_ATTN_QSKIP=0" "B2S_ENC_OVERLAP=0"; do
  ms=$(env $arm python bench.py --no-cpu-baseline --no-roofline-pass --no-extras --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
  echo "[$arm] $ms" | tee -a $out/r3y_ab.txt
done; done
