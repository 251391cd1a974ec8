#!/bin/bash
# Builds libptamd.so (HIP kernels + C ABI + host facade) for gfx950, in-tree.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
  -Wall -Wno-unused-function -Wno-unused-variable \
  -o ${PTAMD_OUT:-libptamd.so} csrc/ptamd.hip host/hmath.cpp host/bvh.cpp host/scene.cpp host/host_c_api.cpp host/offscreen.cpp host/scene_loader.cpp -ldl "$@"
