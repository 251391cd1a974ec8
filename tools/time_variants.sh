#!/bin/bash
# Times the closest-hit query with every library variant given (paths relative to the package directory).
cd "$(dirname "$0")/.."
for lib in "$@"; do
  PTAMD_LIB=$PWD/cuda-path-tracer_amd/$lib timeout -k 10 120 python -u tools/mode2_time.py ${SCENE:-helmet} 2>&1 | tail -n 1 || exit 1
done
