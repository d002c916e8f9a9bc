#!/bin/bash
# HIP API calls of the host around the end of a training step (where the main stream runs dry): name, start (us from the step's first kernel), duration.
repo=$PWD; out=$repo/gpurun_out/hostapi; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $out -o t -- python $repo/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-extras > $out/log.txt 2>&1
cd $repo
python - <<'PY'
import csv, glob
out = "gpurun_out/hostapi"
kt = list(csv.DictReader(open(glob.glob(out + "/*kernel_trace.csv")[0])))
api = list(csv.DictReader(open(glob.glob(out + "/*hip_api_trace.csv")[0])))
kt.sort(key=lambda r: int(r["Start_Timestamp"]))
steps = [i for i, r in enumerate(kt) if "k_embed_prep_fwd" in r["Kernel_Name"]]
t0 = int(kt[steps[-2]]["Start_Timestamp"]); t1 = int(kt[steps[-1]]["Start_Timestamp"])
api.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in api if t0 <= int(r["Start_Timestamp"]) < t1]
print("api calls in the step:", len(sel))
from collections import Counter, defaultdict
tot = defaultdict(float); cnt = Counter()
for r in sel:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot[r["Function"]] += d; cnt[r["Function"]] += 1
for f, v in sorted(tot.items(), key=lambda kv: -kv[1])[:12]:
    print("%-40s n=%4d total %8.1f us  avg %6.2f" % (f, cnt[f], v, v / cnt[f]))
# calls longer than 20 us, and the idle time between calls longer than 30 us (host busy outside HIP)
prev_end = None
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev_end is not None and (s - prev_end) / 1e3 > 30:
        print("  host outside HIP for %6.1f us before %s at %8.1f" % ((s - prev_end) / 1e3, r["Function"], (s - t0) / 1e3))
    if (e - s) / 1e3 > 20:
        print("  long call %-32s %7.1f us at %8.1f" % (r["Function"], (e - s) / 1e3, (s - t0) / 1e3))
    prev_end = e
PY
rm -rf $out
