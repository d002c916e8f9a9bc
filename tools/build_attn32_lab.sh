#!/bin/bash
# Builds tools/bin/attn32_lab -- see tools/attn32_lab.hip
set -e
cd "$(dirname "$0")"
mkdir -p bin
C=../few-shot-transformer-tts_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -DB2S_LAB"
hipcc $F -c $C/attention.hip -o bin/lab_attention.o &
hipcc $F -c $C/attention32.hip -o bin/lab_attention32.o &
hipcc $F -c attn32_lab.hip -o bin/lab_attn32_main.o &
wait
hipcc --offload-arch=gfx950 bin/lab_attn32_main.o bin/lab_attention.o bin/lab_attention32.o -o bin/attn32_lab
echo built
