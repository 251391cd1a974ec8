#!/usr/bin/env python3
"""Prints per-kernel means of the counters collected by tools/profile_sq.sh."""
import collections
import csv
import glob
import os
import sys

kernel = sys.argv[1] if len(sys.argv) > 1 else "k_trace_closest<false, false>"
base = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/sq"
vals = {}
for name in "abc":
    fs = glob.glob(f"{base}/{name}/vm/*_counter_collection.csv")
    if not fs:
        continue
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        if kernel not in r["Kernel_Name"]:
            continue
        d[r["Counter_Name"]][0] += 1
        d[r["Counter_Name"]][1] += float(r["Counter_Value"])
        vals["VGPR"] = r.get("VGPR_Count"); vals["LDS"] = r.get("LDS_Block_Size"); vals["grid"] = r.get("Grid_Size")
    for k, (n, v) in d.items():
        vals[k] = v / n
for k, v in vals.items():
    print(f"{k:34s} {v}")
if "SQ_THREAD_CYCLES_VALU" in vals and "SQ_ACTIVE_INST_VALU" in vals:
    print("lane utilisation of VALU instructions: %.3f" % (vals["SQ_THREAD_CYCLES_VALU"] / (64 * vals["SQ_ACTIVE_INST_VALU"])))
if "SQ_INSTS_VALU" in vals and "SQ_BUSY_CYCLES" in vals:
    print("VALU instructions x 4 cycles over SIMD time (1024 SIMDs, 32 SEs): %.3f" % (vals["SQ_INSTS_VALU"] * 4 / 1024 / (vals["SQ_BUSY_CYCLES"] / 32)))
if "SQ_ACTIVE_INST_VALU" in vals and "SQ_BUSY_CYCLES" in vals:
    print("VALU pipe busy (SQ_ACTIVE_INST_VALU quad-cycles x 4 over SIMD time): %.3f" % (vals["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (vals["SQ_BUSY_CYCLES"] / 32)))
