#!/usr/bin/env python3
"""Condenses gpurun_out/prof (written by tools/profile.sh) into the files kept under profiles/:
  <tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the default bench command
  <tag>_pmc_per_kernel.csv       mean counter value per dispatch, per kernel, of every --pmc pass
  traffic.json                   memory-side bytes per k_trace_closest launch and, when tools/profile_sq.sh has run,
                                 the shader-engine counters that name what limits it (read by bench.py)
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
    """Kernel name with its template arguments: k_trace_closest<false, false> is the reference-exact
    query, <false, true> the 4-wide one, <true, ...> the counting variants."""
    return name.split("(")[0].replace("void ", "").replace("ptd::", "").strip()


EXACT = "k_trace_closest<false, false>"
WIDE = "k_trace_closest<false, true>"


def sq_binding(base, kernel=EXACT):
    """What limits a closest-hit kernel, from the shader-engine counter passes of tools/profile_sq.sh."""
    vals = {}
    for name in "abc":
        f = newest(os.path.join(base, name, "*", "*counter_collection.csv"))
        if not f:
            continue
        acc = defaultdict(list)
        for r in csv.DictReader(open(f)):
            if short(r["Kernel_Name"]) != kernel:
                continue
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            vals["vgpr_count_field"] = r.get("VGPR_Count")
            vals["lds_bytes_per_workgroup"] = r.get("LDS_Block_Size")
        for k, v in acc.items():
            vals[k] = sum(v) / len(v)
    if "SQ_ACTIVE_INST_VALU" not in vals or "SQ_BUSY_CYCLES" not in vals:
        return None
    simds, ses = 1024.0, 32.0
    kernel_cycles = vals["SQ_BUSY_CYCLES"] / ses  # SQ_BUSY_CYCLES is summed over the 32 shader engines
    out = {"counters": {k: v for k, v in vals.items() if isinstance(v, float)},
           "kernel_cycles": kernel_cycles,
           # SQ_ACTIVE_INST_VALU counts quad-cycles in which a SIMD executes a vector instruction: a direct
           # measure of VALU pipe occupancy that assumes no cycles-per-instruction figure
           "valu_busy_frac": vals["SQ_ACTIVE_INST_VALU"] * 4.0 / (simds * kernel_cycles),
           "valu_quad_cycles_per_instruction": vals["SQ_ACTIVE_INST_VALU"] / vals["SQ_INSTS_VALU"] if vals.get("SQ_INSTS_VALU") else None,
           "lane_utilisation": vals["SQ_THREAD_CYCLES_VALU"] / (64.0 * vals["SQ_ACTIVE_INST_VALU"]) if vals.get("SQ_THREAD_CYCLES_VALU") else None,
           "wave_cycles_waiting_frac": vals["SQ_WAIT_ANY"] / vals["SQ_WAVE_CYCLES"] if vals.get("SQ_WAIT_ANY") and vals.get("SQ_WAVE_CYCLES") else None}
    if vals.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
        out["tcp_accesses_per_cycle_per_cu"] = vals["TCP_TOTAL_CACHE_ACCESSES_sum"] / (256.0 * kernel_cycles)
        out["l1_hit_frac"] = 1.0 - vals.get("TCP_TCC_READ_REQ_sum", 0.0) / vals["TCP_TOTAL_CACHE_ACCESSES_sum"]
    out["resource"] = ("waves in flight: 16 per CU is all that 116-128 registers and 10 KB of LDS per wave allow (about 620 B of on-chip state per "
                       "ray, 95 % of a CU's register file + LDS), and with 4 waves per SIMD neither pipe is full: the vector ALUs are busy "
                       "valu_busy_frac of the launch with every instruction priced at 4 cycles (measured per class, profiles/r2_ubench_*.txt: 2.7 "
                       "for fp32 add / mul / fma, 4 to 4.7 for min / max / compare / select / integer) at lane_utilisation of the lanes, the L1 "
                       "takes tcp_accesses_per_cycle_per_cu lane requests against 1.7 measured for dependent gathers.  The launch time follows the "
                       "wave count (8 / 12 / 16 waves per CU: 2.54 / 1.99 / 1.68 ms) and the length of a round: every slab test evaluated twice "
                       "costs 9-13 %, every node piece requested twice 30-46 %, a third of the node visits served from LDS nothing, because a round "
                       "waits for its slowest lane (profiles/r2_load_vs_valu_sensitivity.txt).  HBM carries a few per cent of the algorithmic bytes")
    out["source"] = "tools/profile_sq.sh: rocprofv3 --pmc, three separate passes over bench.py --steps 3 --spp-steps 0"
    return out


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
    s32, g64 = means[("calib_fetch", "stream32", "FETCH_SIZE")], means[("calib_fetch", "gather64", "FETCH_SIZE")]
    w32 = means[("calib_write", "stream32", "WRITE_SIZE")]
    records = {}
    for key, kernel in (("k_trace_closest", EXACT), ("k_trace_closest_wide", WIDE)):
        if ("fetch", kernel, "FETCH_SIZE") not in means:
            continue
        fetch_kb = means[("fetch", kernel, "FETCH_SIZE")]
        write_kb = means[("write", kernel, "WRITE_SIZE")]
        traffic = fetch_kb * 1024 + rays * 32 * 0.5 + write_kb * 1024
        records[key] = {
            "kernel": kernel, "workload": bench["config"]["workload"], "rays": rays, "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
            "correction": f"FETCH_SIZE counts a wide coalesced stream at 1/2 (pmc_calib stream32: {s32:.0f} KB for 536.9 MB) and random 64-byte "
                          f"records exactly (gather64: {g64:.0f} KB for 1073.7 MB); the 32-byte ray stream (rays*32 B) is therefore added once more "
                          f"at half weight; WRITE_SIZE is exact for 32-byte records ({w32:.0f} KB for 536.9 MB)",
            "traffic_bytes": traffic,
            "l2_hit_frac": (means[("l2", kernel, "TCC_HIT_sum")] / (means[("l2", kernel, "TCC_HIT_sum")] + means[("l2", kernel, "TCC_MISS_sum")])
                            if ("l2", kernel, "TCC_HIT_sum") in means else None),
            "binding": sq_binding(os.path.join(ROOT, "gpurun_out", "sq"), kernel),
            "source": f"profiles/{tag}_pmc_per_kernel.csv (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of bench.py --steps 3)"}
    json.dump(records, open(os.path.join(OUT, "traffic.json"), "w"), indent=1)
    traffic = records["k_trace_closest"]["traffic_bytes"]
    print("traffic bytes per launch", traffic)
    for r in rows:
        if r[1] == EXACT:
            print(r)


if __name__ == "__main__":
    main()
