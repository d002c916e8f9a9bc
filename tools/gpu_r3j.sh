#!/bin/bash
out=$PWD/gpurun_out; : > $out/r3j_ab.txt
for r in 1 2; do
for arm in "B2S_DW_STAGES=2" "B2S_DW_STAGES=4" "B2S_DW_STAGES=8" "B2S_DW_STAGES=100"; do
  ms=$(env $arm python bench.py --no-cpu-baseline --no-roofline-pass --no-extras --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
  echo "[$arm] $ms" | tee -a $out/r3j_ab.txt
done; done
