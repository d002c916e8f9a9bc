#!/bin/bash
# kernel trace of the encoder-chain experiment.  usage: bash tools/enc_chain_trace.sh <nchains>
n=${1:-2}
repo=$PWD; out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
ENC_CHAINS=$n rocprofv3 --kernel-trace --output-format csv -d $out/encc$n -o t -- python $repo/tools/enc_chain_lab.py > $out/encc$n.log 2>&1
cd $repo
find $out/encc$n -name "*.db" -delete
python - <<PY
import csv, glob
f = glob.glob("$out/encc$n/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-400:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[:120]:
    print("%9.1f %7.1f q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
PY
rm -f $out/encc$n/*kernel_trace.csv
