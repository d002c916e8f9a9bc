#!/bin/bash
# rocprofv3 kernel trace (timestamps kept) of the training step, default two-stream schedule and all-on-one-stream.
# Usage (through gpurun): bash tools/gpu_trace.sh <tag> [extra bench args]
tag=${1:-tr}; shift
repo=$PWD
out=$repo/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python bench.py --no-cpu-baseline --no-extras > $out/${tag}_bench.json 2> $out/${tag}_bench.err; tail -c 300 $out/${tag}_bench.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_two -o t -- python $repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-extras "$@" > $out/${tag}_two.log 2>&1
B2S_DW_GROUP_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_one -o t -- python $repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-extras "$@" > $out/${tag}_one.log 2>&1
cd $repo
find $out/${tag}_two $out/${tag}_one -name "*.db" -delete
tail -2 $out/${tag}_two.log $out/${tag}_one.log
