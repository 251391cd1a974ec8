#!/bin/bash
# Samples per second in every trace mode with every library variant given (paths relative to the package directory).
cd "$(dirname "$0")/.."
for lib in "$@"; do
  echo "== $lib"
  PTAMD_LIB=$PWD/cuda-path-tracer_amd/$lib timeout -k 10 200 python -u tools/spp_modes.py ${SCENE:-helmet} 2>&1 | grep -v amdgpu.ids || exit 1
done
