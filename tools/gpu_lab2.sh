#!/bin/bash
out=gpurun_out/lab_r2b.txt; : > $out
S="8148,768,64,0,0;8148,768,768,0,0;8148,768,3072,0,0;8148,2304,768,0,0;8148,3072,768,0,1;1596,512,512,0,0"
for rep in 1 2; do
for b in gemm_lab gemm_lab_sc1; do
  echo "== $b plain" >> $out; LAB_ROT=4 LAB_SHAPES="$S" tools/bin/$b 40 >> $out 2>&1
  echo "== $b epi" >> $out; LAB_EPI=1 LAB_ROT=4 LAB_SHAPES="$S" tools/bin/$b 40 >> $out 2>&1
done
done
for b in stamp_lab stamp_lab_sc1; do echo "== $b" >> $out; tools/bin/$b 8148 768 768 0 0 0 >> $out; tools/bin/$b 8148 768 768 0 0 1 >> $out; done
grep -v "^  WG\|mix" $out
