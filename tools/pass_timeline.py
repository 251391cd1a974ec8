#!/usr/bin/env python3
"""Prints the kernel timeline of the last render pass found in a rocprofv3 kernel trace."""
import csv
import glob
import os
import sys

f = max(glob.glob((sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof/kt") + "/vm/*_kernel_trace.csv"), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_film" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["End_Timestamp"])
print("pass wall ms", (int(rows[b]["End_Timestamp"]) - t0) / 1e6)
for r in rows[a + 1 : b + 1]:
    n = r["Kernel_Name"].split("(")[0][:24]
    print(f"{n:24s} start {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us")
