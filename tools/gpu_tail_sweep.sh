#!/bin/bash
# lab sweep of the tail policy (weight-gradient groups held for the encoder backward, their tile cap, the tail optimizer grid)
repo=$PWD; out=$repo/gpurun_out; export B2S_LIB_PATH=$repo/tools/bin/libb2s_hip_lab.so
run() { env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline-pass --no-extras 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2; do
for arm in "B2S_LAB_DW_TAIL_LAYERS=2" "B2S_LAB_DW_TAIL_LAYERS=1 B2S_LAB_DW_TAIL_CAP=200" "B2S_LAB_DW_TAIL_LAYERS=1 B2S_LAB_DW_TAIL_CAP=176" "B2S_LAB_DW_TAIL_LAYERS=1 B2S_LAB_DW_TAIL_CAP=224" "B2S_LAB_DW_TAIL_LAYERS=1 B2S_LAB_DW_TAIL_CAP=256" "B2S_LAB_DW_TAIL_LAYERS=1 B2S_LAB_TAIL_ADAM_WG=384" "B2S_LAB_DW_TAIL_LAYERS=1 B2S_LAB_TAIL_ADAM_WG=640"; do
  echo "$arm: $(run $arm)"
done; done | tee $out/r5_tail_sweep.txt
