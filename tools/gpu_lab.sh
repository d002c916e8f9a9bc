#!/bin/bash
# scratch GPU lab call (edited per experiment)
mkdir -p gpurun_out
repo=$PWD; out=$repo/gpurun_out
export TMPDIR=/tmp
cd /tmp
for arm in 0 1; do
B2S_RESIDUAL_BF16=$arm rocprofv3 --kernel-trace --stats --output-format csv -d $out/lab_x$arm -o t -- python $repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-extras > $out/lab_x$arm.log 2>&1
find $out/lab_x$arm -name "*kernel_trace.csv" -delete; find $out/lab_x$arm -name "*.db" -delete
done
cd $repo
bash tools/gpu_ab.sh lab 4 "B2S_RESIDUAL_BF16=0" "-"
