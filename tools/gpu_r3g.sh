#!/bin/bash
# BN kernels (fused statistics, vectorised backward), vectorised loss: parity then A/B against the previous library
out=$PWD/gpurun_out; mkdir -p $out
python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_ops.py tests/test_gpu_extensions.py tests/test_gpu_trainer_state.py -q -m gpu -x > $out/r3g_tests.log 2>&1; tail -4 $out/r3g_tests.log
: > $out/r3g_ab.txt
for r in 1 2; do
for arm in "B2S_LIB_PATH=$PWD/tools/bin/libb2s_epi.so" "B2S_BN_SEPARATE=1 B2S_BN_SCALAR=1" "B2S_X=0"; do
  ms=$(env $arm python bench.py --no-cpu-baseline --no-roofline-pass --no-extras --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
  echo "[$arm] $ms" | tee -a $out/r3g_ab.txt
done; done
