#!/bin/bash
out=gpurun_out/lab_r2c.txt; : > $out
S="8148,2304,768,0,0;8148,3072,768,0,0;8148,3072,768,0,1;8148,768,768,0,0;8148,768,3072,0,0;4096,4096,4096,0,0"
for rep in 1 2; do
for v in 0 1; do
  echo "== 2WG=$v plain" >> $out; B2S_GEMM_2WG=$v LAB_ROT=4 LAB_SHAPES="$S" tools/bin/gemm_lab 40 >> $out 2>&1
  echo "== 2WG=$v epi" >> $out; B2S_GEMM_2WG=$v LAB_EPI=1 LAB_ROT=4 LAB_SHAPES="$S" tools/bin/gemm_lab 40 >> $out 2>&1
done
done
grep -v "mix" $out
