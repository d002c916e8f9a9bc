#!/bin/bash
# encoder on its own stream: parity, then A/B
out=$PWD/gpurun_out; mkdir -p $out
python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_extensions.py tests/test_gpu_trainer_state.py tests/test_gpu_edge_dp.py tests/test_gpu_checkpoint.py -q -m gpu -x > $out/r3h_tests.log 2>&1; tail -4 $out/r3h_tests.log
: > $out/r3h_ab.txt
for r in 1 2 3; do
for arm in "B2S_ENC_OVERLAP=0" "B2S_ENC_OVERLAP=1"; do
  ms=$(env $arm python bench.py --no-cpu-baseline --no-roofline-pass --no-extras --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
  echo "[$arm] $ms" | tee -a $out/r3h_ab.txt
done; done
