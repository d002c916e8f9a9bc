#!/bin/bash
repo=$PWD; out=$repo/gpurun_out; export TMPDIR=/tmp; : > $out/r3r.txt
cd /tmp
for rows in 32 48 64 96; do
  rm -rf /tmp/lnp; B2S_LN_BWD_ROWS=$rows rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lnp -o ln -- python $repo/tools/ln_rate.py > /dev/null 2>&1
  python - <<PY >> $out/r3r.txt
import csv,glob
f=glob.glob('/tmp/lnp/**/ln_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_ln_bwd' in r['Name']: print("rows/wg $rows: bwd %.2f us" % (float(r['AverageNs'])/1e3))
PY
done
cd $repo
for r in 1 2; do
for arm in "B2S_LN_BWD_ROWS=12" "B2S_LN_BWD_ROWS=24" "B2S_LN_BWD_ROWS=32" "B2S_LN_BWD_ROWS=48"; do
  ms=$(env $arm python bench.py --no-cpu-baseline --no-roofline-pass --no-extras --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
  echo "[$arm] $ms" | tee -a $out/r3r.txt
done; done
cat $out/r3r.txt
