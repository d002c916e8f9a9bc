#!/bin/bash
# kernel-trace timeline of the training step (fused encoder), plus the same with the unfused encoder
repo=$PWD; out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for arm in fused unfused; do
  envs=""; [ $arm = unfused ] && envs="B2S_ENC_FUSED=0"
  env $envs rocprofv3 --kernel-trace --stats --output-format csv -d $out/r4d_$arm -o t -- python $repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-extras > $out/r4d_$arm.log 2>&1
  f=$(find $out/r4d_$arm -name "*kernel_trace.csv" | head -1)
  python $repo/tools/timeline.py $f --list > $out/r4d_timeline_$arm.txt 2>&1
  find $out/r4d_$arm -name "*.db" -delete; find $out/r4d_$arm -name "*kernel_trace.csv" -delete
done
