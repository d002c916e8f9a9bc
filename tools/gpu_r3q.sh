#!/bin/bash
out=$PWD/gpurun_out; : > $out/r3q_ab.txt
for r in 1 2; do
for arm in "B2S_LIB_PATH=$PWD/tools/bin/libb2s_lnw3.so B2S_LN_BWD_ROWS=12" "B2S_LN_BWD_ROWS=12" "B2S_LN_BWD_ROWS=8" "B2S_LIB_PATH=$PWD/tools/bin/libb2s_lnw3.so B2S_LN_BWD_ROWS=8"; do
  ms=$(env $arm python bench.py --no-cpu-baseline --no-roofline-pass --no-extras --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
  echo "[$arm] $ms" | tee -a $out/r3q_ab.txt
done; done
