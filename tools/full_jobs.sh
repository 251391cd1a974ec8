#!/bin/bash
# The reference's configurations as whole jobs on one GPU (scene load, BVH build on the device,
# render, final PNG): wall time and samples per second.  Output under gpurun_out/jobs/.
set -e -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out/jobs
mkdir -p $OUT
run() { echo "== $*"; local t0=$(date +%s.%N); timeout -k 10 500 python3 -u tools/render.py "$@" --device-bvh 2>&1 | tail -1; python3 -c "import time,sys; print(\"process wall %.1f s\" % (time.time() - float(sys.argv[1])))" $t0; }
run cornell --width 512 --height 512 --depth 8 --iterations 64 --out $OUT/cornell
run pbr_metallic --width 1920 --height 1080 --depth 8 --iterations 256 --out $OUT/pbr_metallic
run pbr_ior --width 1920 --height 1080 --depth 8 --iterations 256 --out $OUT/pbr_ior
run teapot --width 1920 --height 1080 --iterations 256 --out $OUT/teapot_depth100
run teapot --width 1920 --height 1080 --depth 8 --iterations 256 --out $OUT/teapot_depth8
run helmet --iterations 1024 --out $OUT/helmet
run instancing_stress --width 3840 --height 2160 --depth 8 --iterations 1024 --batch 32 --out $OUT/party_standin_4k
ls -la $OUT
