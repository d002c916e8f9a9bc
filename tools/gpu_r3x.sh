#!/bin/bash
out=$PWD/gpurun_out; : > $out/r3x_ab.txt
python -m pytest tests -q -m gpu -x > $out/r3x_tests.log 2>&1; tail -3 $out/r3x_tests.log | cut -c1-200
for r in 1 2 3; do
for arm in "B2S_DW_TAIL_LAYERS=0" "B2S_DW_TAIL_LAYERS=2 B2S_DW_TAIL_CAP=200" "B2S_DW_TAIL_LAYERS=2 B2S_DW_TAIL_CAP=100000" "B2S_DW_TAIL_LAYERS=2 B2S_DW_TAIL_CAP=216"; do
  ms=$(env $arm python bench.py --no-cpu-baseline --no-roofline-pass --no-extras --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
  echo "[$arm] $ms" | tee -a $out/r3x_ab.txt
done; done
