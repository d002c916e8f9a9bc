#!/bin/bash
# scratch GPU lab call (edited per experiment)
mkdir -p gpurun_out
bash tools/gpu_pmc.sh lab_new > /dev/null 2>&1
B2S_DW_XCD_ORDER=0 bash tools/gpu_pmc.sh lab_old > /dev/null 2>&1
ls gpurun_out | head -30
