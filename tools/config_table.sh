#!/bin/bash
# One bench line per configuration of the reference's benchmark list (scene, resolution), each in
# its own bounded process; the script stops at the first failure.  Results: gpurun_out/configs/.
set -e -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out/configs
mkdir -p $OUT
for scene in cornell pbr_metallic pbr_ior teapot helmet instancing_stress; do
  timeout -k 10 300 python3 -u bench.py --scene $scene --cpu-seconds 0 --spp-steps 1024 > $OUT/$scene.json 2> $OUT/$scene.err
  tail -1 $OUT/$scene.json | cut -c1-200
done
for scene in helmet instancing_stress; do
  timeout -k 10 300 python3 -u bench.py --scene $scene --width 3840 --height 2160 --cpu-seconds 0 --spp-steps 256 --spp-batch 32 > $OUT/${scene}_4k.json 2> $OUT/${scene}_4k.err
  tail -1 $OUT/${scene}_4k.json | cut -c1-200
done
