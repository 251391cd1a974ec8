#!/usr/bin/env python3
"""Condenses gpurun_out/prof (written by tools/profile.sh) into the files kept under profiles/:
  <tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the default bench command
  <tag>_pmc_per_kernel.csv       mean counter value per dispatch, per kernel, of every --pmc pass
  traffic.json                   memory-side bytes per k_trace_closest launch (read by bench.py)
Usage: python tools/summarize_profile.py <tag>"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "gpurun_out", "prof")
OUT = os.path.join(ROOT, "profiles")


def newest(pattern):
    files = glob.glob(pattern)
    return max(files, key=os.path.getmtime) if files else None


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("ptd::", "")
    return n.split("<")[0] + ("<true>" if "<true>" in n else "")


def main():
    tag = sys.argv[1]
    shutil.copy(newest(os.path.join(PROF, "kt", "*", "*kernel_stats.csv")), os.path.join(OUT, f"{tag}_bench_kernel_stats.csv"))
    rows = []
    means = {}
    for p in ("calib_fetch", "calib_write", "fetch", "write", "l2"):
        f = newest(os.path.join(PROF, p, "*", "*counter_collection.csv"))
        if not f:
            continue
        acc = defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            if k.startswith("__amd"):
                continue
            rows.append((p, k, c, len(v), sum(v) / len(v)))
            means[(p, k, c)] = sum(v) / len(v)
    with open(os.path.join(OUT, f"{tag}_pmc_per_kernel.csv"), "w") as f:
        f.write("pass,kernel,counter,dispatches,mean_per_dispatch\n")
        for r in rows:
            f.write(f"{r[0]},{r[1]},{r[2]},{r[3]},{r[4]:.1f}\n")
    line = [ln for ln in open(os.path.join(PROF, "bench_fetch.log")) if ln.startswith("{")][-1]
    bench = json.loads(line)
    rays = bench["config"]["rays_per_rank"]
    fetch_kb = means[("fetch", "k_trace_closest", "FETCH_SIZE")]
    write_kb = means[("write", "k_trace_closest", "WRITE_SIZE")]
    s32, g64 = means[("calib_fetch", "stream32", "FETCH_SIZE")], means[("calib_fetch", "gather64", "FETCH_SIZE")]
    w32 = means[("calib_write", "stream32", "WRITE_SIZE")]
    traffic = fetch_kb * 1024 + rays * 32 * 0.5 + write_kb * 1024
    json.dump({"k_trace_closest": {
        "workload": bench["config"]["workload"], "rays": rays, "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
        "correction": f"FETCH_SIZE counts a wide coalesced stream at 1/2 (pmc_calib stream32: {s32:.0f} KB for 536.9 MB) and random 64-byte "
                      f"records exactly (gather64: {g64:.0f} KB for 1073.7 MB); the 32-byte ray stream (rays*32 B) is therefore added once more "
                      f"at half weight; WRITE_SIZE is exact for 32-byte records ({w32:.0f} KB for 536.9 MB)",
        "traffic_bytes": traffic,
        "source": f"profiles/{tag}_pmc_per_kernel.csv (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of bench.py --steps 3)"}},
        open(os.path.join(OUT, "traffic.json"), "w"), indent=1)
    print("traffic bytes per launch", traffic)
    for r in rows:
        if r[1].startswith("k_trace_closest") and "<true>" not in r[1]:
            print(r)


if __name__ == "__main__":
    main()
