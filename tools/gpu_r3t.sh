#!/bin/bash
out=$PWD/gpurun_out; : > $out/r3t_ab.txt
python -m pytest tests -q -m gpu -x -k "trainer or lj_shape or finetune or postnet or loss or packer or abandoned or edge" > $out/r3t_tests.log 2>&1; tail -3 $out/r3t_tests.log
for r in 1 2 3; do
for arm in "B2S_ZERO_OVERLAP=1" "B2S_ZERO_OVERLAP=0" "B2S_ZERO_OVERLAP=0 B2S_CONV_REDUCE_V1=1"; do
  ms=$(env $arm python bench.py --no-cpu-baseline --no-roofline-pass --no-extras --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
  echo "[$arm] $ms" | tee -a $out/r3t_ab.txt
done; done
