#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs (separate passes) into per-kernel per-launch HBM traffic.

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [mfma_counter_collection.csv]
The optional third pass (--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE) adds mfma_util = MFMA-busy cycles summed over the
chip's 1024 SIMDs / (1024 x GRBM_GUI_ACTIVE per XCD), i.e. the fraction of matrix-pipe cycles a kernel keeps busy (the rocprofv3
derived metric MfmaUtil).  rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs' counter instances: it is divided by 8 here
(cross-check: the GEMM kernels' mfma_util then equals their TFLOP/s over the 2.5 PFLOP/s peak).
FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts 128-byte requests at 64 bytes, so it is doubled
(MI355X_MICROARCH.md, section HBM); WRITE_SIZE is reported as is (uncalibrated per that guide).
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*$", "", name)            # drop the argument list
    return name.strip()


def load(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    mb, ga = (load(sys.argv[4], "SQ_VALU_MFMA_BUSY_CYCLES"), load(sys.argv[4], "GRBM_GUI_ACTIVE")) if len(sys.argv) > 4 else ({}, {})
    rows = []
    for k in f:
        fv, wv = f[k], w.get(k, [])
        rows.append({"kernel": k, "launches": len(fv),
                     "FETCH_SIZE_KB_per_launch": round(sum(fv) / len(fv), 1),
                     "fetch_MB_per_launch_corrected_x2": round(2 * sum(fv) / len(fv) / 1024, 2),
                     "WRITE_SIZE_KB_per_launch": round(sum(wv) / len(wv), 1) if wv else None,
                     "total_fetch_MB_corrected": round(2 * sum(fv) / 1024, 1),
                     "mfma_util": round(sum(mb[k]) / (1024.0 * sum(ga[k]) / 8.0), 4) if k in mb and k in ga and sum(ga[k]) > 0 else None})
    rows.sort(key=lambda r: -r["total_fetch_MB_corrected"])
    import os
    # the tree the counters were collected on (B2S_COMMIT: the GPU box has no .git) -- bench.py copies it into roofline.traffic_source
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench_decode import csrc_sha16
    json.dump({"commit": os.environ.get("B2S_COMMIT", "unknown"), "csrc_sha16": csrc_sha16(), "tool": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, "
               "separate passes (tools/gpu_pmc.sh)", "rows": rows}, open(out, "w"), indent=1)
    for r in rows[:12]:
        print("%-60s launches %5d  fetch %9.2f MB/launch  write %9.2f MB/launch  mfma_util %s" % (
            r["kernel"][:60], r["launches"], r["fetch_MB_per_launch_corrected_x2"], (r["WRITE_SIZE_KB_per_launch"] or 0) / 1024, r["mfma_util"]))


if __name__ == "__main__":
    main()
