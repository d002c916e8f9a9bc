#!/bin/bash
repo=$PWD; out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for arm in new prev; do
  envs=""; [ $arm = prev ] && envs="B2S_LIB_PATH=$repo/tools/bin/libb2s_prev.so"
  env $envs rocprofv3 --kernel-trace --stats --output-format csv -d $out/r4s_$arm -o t -- python $repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-extras > $out/r4s_$arm.log 2>&1
  find $out/r4s_$arm -name "*.db" -delete; find $out/r4s_$arm -name "*kernel_trace.csv" -delete
done
