#!/bin/bash
# Are the main stream's idle gaps host-bound?  HIP API trace + kernel trace of a few steps; for every kernel: host launch-call time vs
# the end of the previous kernel on its queue.  usage (through gpurun): bash tools/gpu_hostgap.sh
repo=$PWD; out=$repo/gpurun_out/hostgap; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $out -o t -- python $repo/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-extras > $out/log.txt 2>&1
cd $repo
python - <<'PY'
import csv, glob, re
out = "gpurun_out/hostgap"
kt = list(csv.DictReader(open(glob.glob(out + "/*kernel_trace.csv")[0])))
api = list(csv.DictReader(open(glob.glob(out + "/*hip_api_trace.csv")[0])))
print("api columns:", list(api[0].keys()))
call = {}
for r in api:
    call[r["Correlation_Id"]] = (r["Function"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
kt.sort(key=lambda r: int(r["Start_Timestamp"]))
steps = [i for i, r in enumerate(kt) if "k_embed_prep_fwd" in r["Kernel_Name"]]
seg = kt[steps[-2]:steps[-1]]
t0 = int(seg[0]["Start_Timestamp"])
last_end = {}
rows = []
for r in seg:
    q = r["Queue_Id"]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    c = call.get(r["Correlation_Id"])
    if gap > 5.0 and c:
        # host call finished how long before (+) / after (-) the queue went idle?
        lead = (last_end[q] - c[2]) / 1e3
        rows.append((gap, (s - t0) / 1e3, q, re.sub(r"\(.*", "", r["Kernel_Name"])[-50:], lead, (s - c[2]) / 1e3))
    last_end[q] = max(e, last_end.get(q, 0))
print("gap_us  t_us  queue  kernel  host_call_end_before_queue_idle_us  call_end_to_kernel_start_us")
for g in sorted(rows, reverse=True)[:30]:
    print("%7.1f %8.1f q%s %-50s %9.1f %9.1f" % g)
PY
rm -rf $out
