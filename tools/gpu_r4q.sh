#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_encfused.py -q 2>&1 | tail -12 > gpurun_out/r4q_ops.log
python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_extensions.py tests/test_gpu_dp_race.py -q 2>&1 | tail -12 > gpurun_out/r4q_model.log
bash tools/gpu_ab.sh r4q 3 "B2S_LIB_PATH=$PWD/tools/bin/libb2s_prev.so" "-"
