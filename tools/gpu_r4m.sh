#!/bin/bash
repo=$PWD; out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 MASTER_PORT=29811 B2S_FORCE_DP=1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/r4m_dp -o t -- python $repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-extras > $out/r4m_dp.log 2>&1
f=$(find $out/r4m_dp -name "*kernel_trace.csv" | head -1)
python $repo/tools/timeline.py $f --list > $out/r4m_timeline_dp.txt 2>&1
find $out/r4m_dp -name "*.db" -delete; find $out/r4m_dp -name "*kernel_trace.csv" -delete
