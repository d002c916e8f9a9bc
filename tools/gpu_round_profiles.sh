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
# round 5: the persistent GEMM kernel against the plain one (lab build, B2S_LAB_GEMM_PERSIST = 0 | 1), the step's phases on the trainer's stream
# (unprofiled events), and the step with CUs held during the backward pass at the data-parallel tile policy
if [ -f $repo/tools/bin/libb2s_hip_lab.so ]; then
  for v in 0 1; do echo "B2S_LAB_GEMM_PERSIST=$v"; B2S_LIB_PATH=$repo/tools/bin/libb2s_hip_lab.so B2S_LAB_GEMM_PERSIST=$v timeout 200 python tools/gemm_persist_lab.py 2>&1 | grep "us "; done > $out/${tag}_gemm_persist_ab.txt
  for r in 1 2 3; do for v in 0 1; do echo -n "B2S_LAB_GEMM_PERSIST=$v step ms: "; B2S_LIB_PATH=$repo/tools/bin/libb2s_hip_lab.so B2S_LAB_GEMM_PERSIST=$v timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline-pass --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; done; done >> $out/${tag}_gemm_persist_ab.txt
fi
timeout 200 python tools/step_phases.py 2>&1 | grep "us" > $out/${tag}_step_phases.txt
B2S_GEMM256_NB=4 timeout 300 python tools/cu_loss.py bwd 2>&1 | grep "mode\|held CUs" > $out/${tag}_cu_loss_policy4.txt
for f in $out/${tag}_gemm_persist_ab.txt $out/${tag}_step_phases.txt; do tail -n 3 $f; done
