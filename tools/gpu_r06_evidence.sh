#!/bin/bash
# Round-6 evidence files (profiles/r06_attn32_lab.txt, r06_attn32_pmc.txt, r06_valu_lab.txt, r06_*_ab.txt) in one gpurun call; see tools/README.md
out=gpurun_out
tools/bin/attn32_lab > $out/r06_attn32_lab.txt 2>&1; tail -3 $out/r06_attn32_lab.txt
tools/bin/valu_lab > $out/r06_valu_lab.txt 2>&1; tail -3 $out/r06_valu_lab.txt
rm -f $out/r06_attn32_pmc.txt; bash tools/gpu_attn32_pmc.sh r06_attn32 tools/bin/attn32_lab 0 > /dev/null 2>&1; wc -l $out/r06_attn32_pmc.txt
bash tools/gpu_ab.sh r06_ragged 3 "-" "B2S_COMPACT=0"
bash tools/gpu_ab.sh r06_side_stream 3 "-" "B2S_SIDE_STREAM=0"
L=$PWD/tools/bin/libb2s_hip_lab.so
bash tools/gpu_ab.sh r06_attn32_step 2 "B2S_LIB_PATH=$L B2S_LAB_ATTN32=1" "B2S_LIB_PATH=$L B2S_LAB_ATTN32=0" "B2S_LIB_PATH=$L B2S_LAB_ATTN32=3" "B2S_LIB_PATH=$L B2S_LAB_ATTN32=5" "B2S_LIB_PATH=$L B2S_LAB_ATTN32=9" "B2S_LIB_PATH=$L B2S_LAB_ATTN32=15"
