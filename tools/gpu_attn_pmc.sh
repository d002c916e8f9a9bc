#!/bin/bash
# PMC profile of the attention kernels in the lab harness (separate passes, kernel trace only)
repo=$PWD; out=$repo/gpurun_out; export TMPDIR=/tmp; cd /tmp
rocprofv3 -L > $out/r3_counters.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES SQ_INSTS_VALU_TRANS" \
           "GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_INSTS_SMEM SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/r3_attnpmc_$i -o p -- $repo/tools/bin/attn_lab2 prof > $out/r3_attnpmc_$i.log 2>&1
  f=$(find $out/r3_attnpmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
seen = collections.defaultdict(set)
for r in rows: seen[r["Kernel_Name"][:60]].add(r["Dispatch_Id"])
for k, d in agg.items():
    n = len(seen[k])
    print(k, "launches", n, {c: round(v / n) for c, v in d.items()})
PY
  find $out/r3_attnpmc_$i -name "*.csv" -delete; find $out/r3_attnpmc_$i -name "*.db" -delete
done
