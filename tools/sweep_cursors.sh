#!/bin/bash
# Times the closest-hit leg of the bench for several numbers of work-fetch cursors.
cd "$(dirname "$0")/.."
for g in ${CURSORS:-1 2 4 8 16 32 64}; do
  r=$(PTAMD_QUERY_CURSORS=$g timeout -k 10 120 python3 -u bench.py --steps 20 --warmup 3 --cpu-seconds 0 --spp-steps 0 "$@" 2>&1 | tail -1 | grep -o "kernel_ms\": [0-9.]*\|\"value\": [0-9.]*" | tr '\n' ' ') || exit 1
  echo "cursors $g: $r"
done
