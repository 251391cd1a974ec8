#!/bin/bash
# Shader-engine counters of one batched render pass (tools/queue_sizes.py), per kernel dispatch.
set -e -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/sqpass${PT_TRACE_MODE:-0}
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -- python3 tools/queue_sizes.py 8 > $OUT/a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_FLAT --output-format csv -d $OUT/b -- python3 tools/queue_sizes.py 8 > $OUT/b.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $OUT/c -- python3 tools/queue_sizes.py 8 > $OUT/c.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_ATOMIC_sum --output-format csv -d $OUT/d -- python3 tools/queue_sizes.py 8 > $OUT/d.log 2>&1 || echo "pass d failed"
python3 - <<'PY'
import csv, glob, collections
for name in "abcd":
    fs = glob.glob(f"gpurun_out/sqpass" + __import__("os").environ.get("PT_TRACE_MODE", "0") + f"/{name}/vm/*_counter_collection.csv")
    if not fs: continue
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("ptd::", "")
        rows.setdefault((int(r["Dispatch_Id"]), k), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for (d, k), v in list(sorted(rows.items()))[:8]:
        print(name, d, k, " ".join(f"{a.replace('SQ_','').replace('_sum','')}={b/1e6:.2f}M" for a, b in v.items()))
PY
