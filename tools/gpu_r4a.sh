#!/bin/bash
# round 4, first fused-encoder run: op tests, model tests that touch the encoder at full size, A/B of the fused encoder on the training step
mkdir -p gpurun_out
python -m pytest tests/test_gpu_encfused.py -x -q 2>&1 | tail -15 > gpurun_out/r4a_encf.log
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_model.py -x -q 2>&1 | tail -25 > gpurun_out/r4a_model.log
bash tools/gpu_ab.sh r4a 3 "B2S_ENC_FUSED=0" "-" "B2S_ENC_SLAB_BF16=1"
python tools/tail_lab.py > gpurun_out/r4a_tail.log 2>&1
B2S_ENC_FUSED=0 python tools/tail_lab.py > gpurun_out/r4a_tail_unfused.log 2>&1
