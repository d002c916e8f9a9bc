#!/bin/bash
out=$PWD/gpurun_out; : > $out/r3u_ab.txt
python -m pytest tests -q -m gpu -x -k "trainer or lj_shape or finetune or postnet or packer or edge or modules or forward_loss" > $out/r3u_tests.log 2>&1; tail -3 $out/r3u_tests.log
for r in 1 2 3; do
for arm in "B2S_X=1" "B2S_CONV_DW_GATHER=1"; do
  ms=$(env $arm python bench.py --no-cpu-baseline --no-roofline-pass --no-extras --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
  echo "[$arm] $ms" | tee -a $out/r3u_ab.txt
done; done
