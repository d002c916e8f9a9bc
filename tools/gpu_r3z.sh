#!/bin/bash
out=$PWD/gpurun_out; : > $out/r3z_ab.txt
python -m pytest tests -q -m gpu -x -k "guided or finetune or extension or attention" > $out/r3z_tests.log 2>&1; grep -n "passed\|failed" $out/r3z_tests.log | tail -1
for r in 1 2 3; do
for arm in "B2S_X=0" "B2S_LIB_PATH=$PWD/tools/bin/libb2s_head.so"; do
  ms=$(env $arm python bench.py --mode finetune --no-cpu-baseline --no-roofline-pass --no-extras --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
  echo "[$arm] finetune $ms" | tee -a $out/r3z_ab.txt
done; done
