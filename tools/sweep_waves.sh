#!/bin/bash
# Samples per second of the helmet bench scene for a list of settings of one tuning variable of the library:
#   tools/sweep_waves.sh PTAMD_REFILL_PRIMARY 12 24 32 48 64
set -e -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3
VAR=$1; shift
OUT=gpurun_out/r3/sweep_$VAR.txt
: > $OUT
for v in "$@"; do
  echo "$VAR=$v" >> $OUT
  env $VAR=$v timeout -k 10 120 python3 tools/spp_modes.py helmet --modes ${MODES:-1} --passes 12 2>&1 | grep "spp/s" >> $OUT
done
cat $OUT
