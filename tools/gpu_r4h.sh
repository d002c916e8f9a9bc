#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_model.py tests/test_gpu_extensions.py tests/test_gpu_edge_dp.py -q 2>&1 | tail -25 > gpurun_out/r4h_tests.log
bash tools/gpu_ab.sh r4h 3 "B2S_DX_BF16=0" "-"
