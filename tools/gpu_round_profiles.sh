#!/bin/bash
# Everything the round's profiles/ directory is built from, in one gpurun call.  usage: bash tools/gpu_round_profiles.sh r02
tag=${1:-r02}
repo=$PWD; out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; tail -c 300 $out/${tag}_bench.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_train -o train -- python $repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-extras > $out/${tag}_prof_train.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_decode -o decode -- python $repo/bench.py --mode decode --no-cpu-baseline > $out/${tag}_prof_decode.log 2>&1
cd $repo
find $out/${tag}_prof_train $out/${tag}_prof_decode -name "*.db" -delete
find $out/${tag}_prof_decode -name "*kernel_trace.csv" -delete
bash tools/gpu_pmc.sh ${tag}_pmc
ls $out/${tag}_prof_train $out/${tag}_prof_decode
