#!/bin/bash
repo=$PWD; out=$repo/gpurun_out; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $out/r3i_trace -o t -- python $repo/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-extras > $out/r3i_trace.log 2>&1
cd $repo
f=$(find $out/r3i_trace -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f --list > $out/r3i_timeline.txt 2>&1
tail -62 $out/r3i_timeline.txt
find $out/r3i_trace -name "*.csv" -delete; find $out/r3i_trace -name "*.db" -delete
