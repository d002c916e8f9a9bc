#!/bin/bash
out=$PWD/gpurun_out
for r in 1 2; do
echo "== r02 kernels"; tools/bin/attn_lab_r02
echo "== new kernels RB=1 KBW=1"; B2S_ATTN_RB=1 B2S_ATTN_KBW=1 tools/bin/attn_lab_new
done 2>&1 | tee $out/r3d_attn_old_new.txt
