#!/bin/bash
# Everything the round's profiles/ directory is built from, in one gpurun call.
# usage: B2S_COMMIT=<hash> bash tools/gpu_round_profiles.sh r03      (the GPU box has no .git: the hash comes in through the environment)
tag=${1:-r03}
repo=$PWD; out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; tail -c 300 $out/${tag}_bench.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_train -o train -- python $repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-extras > $out/${tag}_prof_train.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_decode -o decode -- python $repo/bench.py --mode decode --no-cpu-baseline > $out/${tag}_prof_decode.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_ln -o ln -- python $repo/tools/ln_rate.py > $out/${tag}_prof_ln.log 2>&1
cd $repo
python tools/timeline.py $(find $out/${tag}_prof_train -name "*kernel_trace.csv" | head -1) --list > $out/${tag}_train_timeline.txt 2>&1
find $out/${tag}_prof_train $out/${tag}_prof_decode $out/${tag}_prof_ln -name "*.db" -delete
find $out/${tag}_prof_train $out/${tag}_prof_decode $out/${tag}_prof_ln -name "*kernel_trace.csv" -delete
python tools/cu_loss.py > $out/${tag}_cu_loss.txt 2>&1; cat $out/${tag}_cu_loss.txt | tail -7
bash tools/gpu_pmc.sh ${tag}_pmc
bash tools/gpu_pmc_decode.sh ${tag}_pmcd
ls $out/${tag}_prof_train $out/${tag}_prof_decode
