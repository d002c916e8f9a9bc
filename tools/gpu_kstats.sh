#!/bin/bash
# rocprofv3 kernel stats of the training-step benchmark, printed per kernel family: bash tools/gpu_kstats.sh <tag> [env assignments...]
# (through gpurun; writes gpurun_out/<tag>_kernel_stats.csv and prints the rows whose name matches $KS_FILTER, default: all above 0.5 %)
tag=$1; shift
repo=$PWD; out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -o t -- python $repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-extras $KS_ARGS > $out/${tag}_prof.log 2>&1
cd $repo
f=$(find $out/${tag}_prof -name "*kernel_stats.csv" | head -1)
cp $f $out/${tag}_kernel_stats.csv
find $out/${tag}_prof -name "*kernel_trace.csv" -delete; find $out/${tag}_prof -name "*.db" -delete
python - "$out/${tag}_kernel_stats.csv" <<'PY'
import csv, sys, os, re
rows = list(csv.DictReader(open(sys.argv[1])))
flt = os.environ.get("KS_FILTER")
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per step: %.3f ms (13 steps)" % (tot / 13 / 1e6))
for r in rows:
    if (flt and re.search(flt, r["Name"])) or (not flt and float(r["Percentage"]) > 0.5):
        print("%-90s n/step %5.1f  avg %7.1f us  per step %7.1f us" % (r["Name"][:90], int(r["Calls"]) / 13, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 13 / 1e3))
PY
