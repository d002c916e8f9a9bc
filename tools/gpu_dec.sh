#!/bin/bash
# decode: parity tests of the frame loop + bench + kernel stats.  usage: bash tools/gpu_dec.sh <tag> [notest]
tag=$1
r=$PWD; out=$r/gpurun_out
if [ "$2" != "notest" ]; then
python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py -q -m gpu -x -k "decode or pipeline" 2>&1 | tail -3
fi
python bench.py --mode decode --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('decode', d['ms_per_step'], 'ms/frame', d['value'], 'frames/s', d['roofline']['frac'])"
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_dec -o d -- python $r/bench.py --mode decode --no-cpu-baseline > $out/${tag}_dec.log 2>&1
cd $r; find $out/${tag}_dec -name "*.db" -delete; find $out/${tag}_dec -name "*kernel_trace.csv" -delete
head -8 $out/${tag}_dec/d_kernel_stats.csv | cut -c1-140
