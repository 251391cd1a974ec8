#!/bin/bash
# Builds libptamd.so (HIP kernels + C ABI + host facade) for gfx950, in-tree.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
  -Wall -Wno-unused-function -Wno-unused-variable \
  -o ${PTAMD_OUT:-libptamd.so} csrc/ptamd.hip host/hmath.cpp host/bvh.cpp host/scene.cpp host/host_c_api.cpp host/offscreen.cpp host/scene_loader.cpp host/image_decode.cpp -ldl "$@"
# the plain-C example host program (examples/headless_render.c): proves that the headers are C and that a program links against the library alone
${CC:-gcc} -std=c11 -O1 -Wall -Wextra -pedantic -I../include ../examples/headless_render.c -o ../examples/headless_render \
  -L. -lptamd -Wl,-rpath,'$ORIGIN/../cuda-path-tracer_amd' -Wl,-rpath-link,/opt/rocm/lib
