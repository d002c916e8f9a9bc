#!/bin/bash
# usage (on the GPU box): bash tools/gpu_dec_stamps.sh [frames]
B2S_LIB_PATH=$PWD/tools/bin/libb2s_hip_stamped.so python tools/dec_stamps.py ${1:-500} 2>&1 | grep -v amdgpu.ids
