#!/bin/bash
# one rocprofv3 kernel trace (timestamps kept) of the training step.  usage: bash tools/gpu_trace1.sh <tag> [ENV=val ...]
tag=$1; shift
repo=$PWD; out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag} -o t -- python $repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-extras > $out/${tag}.log 2>&1
cd $repo
find $out/${tag} -name "*.db" -delete
tail -c 300 $out/${tag}.log
