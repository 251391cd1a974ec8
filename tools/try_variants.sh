#!/bin/bash
# Times the closest-hit leg of the bench for every diagnostic library variant_*.so.
cd "$(dirname "$0")/.."
for lib in cuda-path-tracer_amd/libptamd.so cuda-path-tracer_amd/variant_*.so; do
  [ -f "$lib" ] || continue
  r=$(PTAMD_LIB=$PWD/$lib timeout -k 10 120 python3 -u bench.py --steps 10 --warmup 2 --cpu-seconds 0 --spp-steps ${SPP_STEPS:-0} 2>&1 | tail -1 | grep -o "kernel_ms\": [0-9.]*\|spp_per_s\": [0-9.]*" | tr '\n' ' ') || exit 1
  echo "$(basename $lib) $r"
done
