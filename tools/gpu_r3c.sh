#!/bin/bash
# round 3: LN fast kernels -- op tests + full-size parity, then A/B of the training step: round-2 library / new with generic LN / new
out=$PWD/gpurun_out; mkdir -p $out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_model.py -q -m gpu -x > $out/r3c_tests.log 2>&1; tail -4 $out/r3c_tests.log
: > $out/r3c_ab.txt
for r in 1 2; do
for arm in "B2S_LIB_PATH=$PWD/tools/bin/libb2s_r02.so" "B2S_LN_GENERIC=1" "B2S_X=0"; do
  ms=$(env $arm python bench.py --no-cpu-baseline --no-roofline-pass --no-extras --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
  echo "[$arm] $ms" | tee -a $out/r3c_ab.txt
done; done
