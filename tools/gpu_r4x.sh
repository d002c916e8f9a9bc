#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_trainer_state.py tests/test_gpu_checkpoint.py tests/test_gpu_edge_dp.py tests/test_gpu_dp_race.py tests/test_gpu_extensions.py -q 2>&1 | tail -8 > gpurun_out/r4x_tests.log
bash tools/gpu_ab.sh r4x 3 "B2S_DW_OVERWRITE=0" "-"
