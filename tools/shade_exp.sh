#!/bin/bash
# Timing experiment: duration of the first k_shade of a batched pass for each diagnostic library.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for lib in cuda-path-tracer_amd/libptamd.so cuda-path-tracer_amd/variant_*.so; do
  rm -rf /tmp/kt && PTAMD_LIB=$PWD/$lib timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 tools/queue_sizes.py 8 > /tmp/kt.log 2>&1 || exit 1
  python3 - "$lib" <<'PY'
import csv, glob, sys
rows = sorted(csv.DictReader(open(glob.glob("/tmp/kt/*/*_kernel_trace.csv")[0])), key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_shade" in r["Kernel_Name"]]
print(sys.argv[1].split("/")[-1], "k_shade us per bounce:", " ".join(f"{x:.0f}" for x in d[:4]))
PY
done
