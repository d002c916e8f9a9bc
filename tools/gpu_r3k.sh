#!/bin/bash
out=$PWD/gpurun_out; : > $out/r3k_ab.txt
B2S_DW_SPLIT=2 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py -q -m gpu -x -k "trainer or lj_shape or finetune" > $out/r3k_tests.log 2>&1; tail -3 $out/r3k_tests.log
for r in 1 2; do
for arm in "B2S_DW_SPLIT=1" "B2S_DW_SPLIT=2" "B2S_DW_SPLIT=3"; do
  ms=$(env $arm python bench.py --no-cpu-baseline --no-roofline-pass --no-extras --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
  echo "[$arm] $ms" | tee -a $out/r3k_ab.txt
done; done
