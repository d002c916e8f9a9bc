#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_extensions.py tests/test_gpu_fullsize.py -q -k "extension or guided or finetune or frozen or ga_" 2>&1 | tail -6 > gpurun_out/r4w_tests.log
for r in 1 2 3; do for arm in "B2S_GA_TABLE=0" "-"; do
  envs=""; [ "$arm" != "-" ] && envs="$arm"
  ms=$(env $envs python bench.py --mode finetune --no-cpu-baseline --no-roofline-pass --no-extras --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
  echo "round $r [$arm] $ms" | tee -a gpurun_out/r4w_ab.txt
done; done
