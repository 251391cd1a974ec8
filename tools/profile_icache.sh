#!/bin/bash
# Instruction-cache counters of one batched render pass, per kernel (first dispatches only).
set -e -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/icache
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/a -- python3 tools/queue_sizes.py 8 > $OUT/a.log 2>&1
python3 - <<'PY'
import csv, glob, collections
rows = collections.OrderedDict()
for r in csv.DictReader(open(glob.glob("gpurun_out/icache/a/vm/*_counter_collection.csv")[0])):
    k = r["Kernel_Name"].split("(")[0].replace("ptd::", "")
    if not k.startswith("k_"): continue
    rows.setdefault((int(r["Dispatch_Id"]), k), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for (d, k), v in list(sorted(rows.items()))[:7]:
    print(d, k, " ".join(f"{a.replace('SQC_','').replace('SQ_','')}={b/1e6:.2f}M" for a, b in v.items()))
PY
