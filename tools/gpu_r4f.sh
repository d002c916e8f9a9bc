#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_encfused.py -x -q 2>&1 | tail -12 > gpurun_out/r4f_encf.log
{ tools/bin/encf_lab 14 114 0 1; tools/bin/encf_lab 14 114 1 1; } > gpurun_out/r4f_stamps.log 2>&1
python tools/encf_lab.py > gpurun_out/r4f_lab.log 2>&1
bash tools/gpu_ab.sh r4f 2 "B2S_ENC_FUSED=0" "-" "B2S_ENC_SLAB_BF16=1"
B2S_ENC_SLAB_BF16=1 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_model.py -x -q 2>&1 | tail -12 > gpurun_out/r4f_model_bf16slab.log
