#!/bin/bash
bash tools/gpu_ab.sh r4e 2 "-" "B2S_LAB_ENC_SKIP=1" "B2S_LAB_ENC_SKIP=2" "B2S_LAB_ENC_SKIP=2 B2S_DW_TAIL_LAYERS=0"
B2S_LAB_ENC_SKIP=2 python tools/tail_lab.py > gpurun_out/r4e_tail_skip2.log 2>&1
