#!/bin/bash
# PMC profile of the attention32 lab (separate passes, kernel trace only).  usage: bash tools/gpu_attn32_pmc.sh <tag> <lab binary> <case index>
tag=$1; bin=$2; cs=$3
repo=$PWD; out=$repo/gpurun_out; export TMPDIR=/tmp; export B2S_LAB_ATTN32=0; cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/${tag}_pmc_$i -o p -- $repo/$bin $cs > $out/${tag}_pmc_$i.log 2>&1
  f=$(find $out/${tag}_pmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' >> $out/${tag}_pmc.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
seen = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); seen[k].add(r["Dispatch_Id"])
for k, d in agg.items():
    n = len(seen[k])
    print(k, "launches", n, {c: round(v / n) for c, v in d.items()})
PY
  rm -rf $out/${tag}_pmc_$i
done
cat $out/${tag}_pmc.txt
