#!/bin/bash
# Sweeps the persistent-grid knobs of the render-pass kernels (spp/s of the bench's second leg).
cd "$(dirname "$0")/.."
run() {
  r=$(env "$@" timeout -k 10 120 python3 -u bench.py --steps 3 --warmup 1 --cpu-seconds 0 --spp-steps 1024 2>&1 | tail -1 | grep -o "spp_per_s\": [0-9.]*") || exit 1
  echo "$* -> $r"
}
for combo in "${@:-12,12,12}"; do
  IFS=, read -r e s a <<< "$combo"
  run PTAMD_EXTEND_WAVES_PER_CU=$e PTAMD_SHADE_WAVES_PER_CU=$s PTAMD_SHADOW_WAVES_PER_CU=$a
done
