#!/bin/bash
out=$PWD/gpurun_out; : > $out/r3o_cu_loss.txt
python tools/cu_loss.py bwd 2>&1 | grep -v amdgpu.ids | tee -a $out/r3o_cu_loss.txt
B2S_GEMM256_NB=4 python tools/cu_loss.py bwd 2>&1 | grep -v amdgpu.ids | tee -a $out/r3o_cu_loss.txt
B2S_GEMM256_NB=4 python tools/cu_loss.py step 2>&1 | grep -v amdgpu.ids | tee -a $out/r3o_cu_loss.txt
