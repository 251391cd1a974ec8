#!/bin/bash
# Sweeps the samples traced per render pass (spp/s of the bench's second leg).
cd "$(dirname "$0")/.."
for b in "${@:-64}"; do
  r=$(timeout -k 10 300 python3 -u bench.py --steps 3 --warmup 1 --cpu-seconds 0 --spp-steps $((b * 12)) --spp-batch $b 2>&1 | tail -1 | grep -o "spp_per_s\": [0-9.]*") || exit 1
  echo "batch $b -> $r"
done
