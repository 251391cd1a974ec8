#!/usr/bin/env python3
"""Condenses the counter passes of tools/profile_pass_sq.sh (one render pass of 8 samples, helmet 1080p, kernels one
after the other) into profiles/r3_sq_counters_render_pass.json: per trace mode and kernel family (k_extend, k_shade,
k_shadow) the counters summed over the pass's dispatches and the two figures bench.py quotes:
  valu_busy_frac    SQ_ACTIVE_INST_VALU quad-cycles x 4 over (1024 SIMDs x kernel cycles), kernel cycles = SQ_BUSY_CYCLES / 32
  lane_utilisation  SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)
Usage: python tools/summarize_pass_sq.py <trace mode> [<trace mode> ...]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles", "r3_sq_counters_render_pass.json")


def family(name):
    base = name.split("(")[0].replace("void ", "").replace("ptd::", "").strip()
    for k in ("k_extend", "k_shade", "k_shadow", "k_generate", "k_film"):
        if base.startswith(k + "<") or base == k:
            return k
    return None


def main():
    try:
        rec = json.load(open(OUT))
    except (OSError, ValueError):
        rec = {}
    for mode in sys.argv[1:]:
        sums = defaultdict(lambda: defaultdict(float))
        dispatches = defaultdict(set)
        for name in "abcd":
            fs = glob.glob(os.path.join(ROOT, "gpurun_out", f"sqpass{mode}", name, "*", "*counter_collection.csv"))
            if not fs:
                continue
            for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
                k = family(r["Kernel_Name"])
                if k is None:
                    continue
                sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
                dispatches[k].add(r["Dispatch_Id"])
        out = {}
        for k, v in sums.items():
            cycles = v.get("SQ_BUSY_CYCLES", 0.0) / 32.0
            line = {"dispatches": len(dispatches[k]), "counters": dict(v)}
            if cycles > 0 and v.get("SQ_ACTIVE_INST_VALU"):
                line["valu_busy_frac"] = v["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cycles)
                line["lane_utilisation"] = v.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * v["SQ_ACTIVE_INST_VALU"])
            if v.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
                line["l1_hit_frac"] = 1.0 - v.get("TCP_TCC_READ_REQ_sum", 0.0) / v["TCP_TOTAL_CACHE_ACCESSES_sum"]
            if v.get("TCC_HIT_sum"):
                line["l2_hit_frac"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v.get("TCC_MISS_sum", 0.0))
            out[k] = line
        rec[f"trace_mode_{mode}"] = out
        for k, line in sorted(out.items()):
            print(mode, k, {a: round(b, 3) for a, b in line.items() if isinstance(b, float)})
    rec["source"] = "tools/profile_pass_sq.sh: rocprofv3 --pmc, four separate passes over tools/queue_sizes.py 8 (one render pass of 8 samples, helmet 1920x1080)"
    json.dump(rec, open(OUT, "w"), indent=1)


if __name__ == "__main__":
    main()
