#!/bin/bash
# Kernel trace of the default bench (no CPU leg), results under gpurun_out/prof/kt.
set -e -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT/kt && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 20 --warmup 3 --cpu-seconds 0 "$@" > $OUT/bench_kt.log 2>&1
grep -o '"value": [0-9.]*\|kernel_ms": [0-9.]*\|spp_per_s": [0-9.]*' $OUT/bench_kt.log
python3 tools/pass_timeline.py $OUT/kt
