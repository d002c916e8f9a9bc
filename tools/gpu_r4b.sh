#!/bin/bash
mkdir -p gpurun_out
python tools/encf_lab.py > gpurun_out/r4b_lab.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4b_prof -o r4b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline-pass --no-extras --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/r4b_prof.log 2>&1
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/r4b_prof | head
f=$(find gpurun_out/r4b_prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
out = open("gpurun_out/r4b_kernel_stats.txt", "w")
for r in rows[:60]:
    out.write("%-110s calls %6s avg %9.1f us total %9.3f ms\n" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
find gpurun_out/r4b_prof -name "*.db" -delete; find gpurun_out/r4b_prof -name "*kernel_trace.csv" -size +30M -delete
