#!/bin/bash
mkdir -p gpurun_out
{ tools/bin/encf_lab 14 114 0 1; tools/bin/encf_lab 14 114 0 0; tools/bin/encf_lab 14 114 1 1; } > gpurun_out/r4c_stamps.log 2>&1
python tools/encf_lab.py > gpurun_out/r4c_lab.log 2>&1
bash tools/gpu_ab.sh r4c 2 "B2S_ENC_FUSED=0" "-" "B2S_DW_TAIL_LAYERS=1" "B2S_DW_TAIL_LAYERS=0" "B2S_DW_TAIL_CAP=128" "B2S_DW_TAIL_LAYERS=1 B2S_DW_TAIL_CAP=128" "B2S_ENC_SLAB_BF16=1"
