#!/bin/bash
# Full measurement pass on the GPU box: parity suite, the three bench modes, rocprofv3 kernel stats.
# Usage (from the repo root, through gpurun):  bash tools/gpu_checkpoint.sh <tag>
tag=${1:-ck}
out=$PWD/gpurun_out
mkdir -p $out
python -m pytest tests -q -m gpu > $out/${tag}_tests.log 2>&1; tail -3 $out/${tag}_tests.log
python bench.py > $out/${tag}_bench_train.json 2> $out/${tag}_bench_train.err; tail -c 600 $out/${tag}_bench_train.json
python bench.py --mode decode > $out/${tag}_bench_decode.json 2> $out/${tag}_bench_decode.err; tail -c 400 $out/${tag}_bench_decode.json
python bench.py --mode finetune --no-cpu-baseline > $out/${tag}_bench_finetune.json 2> $out/${tag}_bench_finetune.err; tail -c 300 $out/${tag}_bench_finetune.json
export TMPDIR=/tmp
repo=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_train -o train -- python $repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-extras > $out/${tag}_prof_train.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_decode -o decode -- python $repo/bench.py --mode decode --no-cpu-baseline > $out/${tag}_prof_decode.log 2>&1
cd $repo
find $out/${tag}_prof_train $out/${tag}_prof_decode -name "*kernel_stats.csv" | head
find $out/${tag}_prof_train $out/${tag}_prof_decode -name "*kernel_trace.csv" -delete
find $out/${tag}_prof_train $out/${tag}_prof_decode -name "*.db" -delete
