#!/bin/bash
# kernel stats of the training step (one stream) with extra hparams.  usage: bash tools/gpu_trace_hp.sh <tag> "<hparams>"
tag=$1; hpar=$2
repo=$PWD; out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
B2S_NO_AUX=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag} -o t -- python $repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-extras --hparams "$hpar" > $out/${tag}.log 2>&1
cd $repo
find $out/${tag} -name "*.db" -delete; find $out/${tag} -name "*kernel_trace.csv" -delete
