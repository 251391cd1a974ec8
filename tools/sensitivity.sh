#!/bin/bash
# Which resource does the closest-hit kernel respond to?  Builds the library with every node requested twice
# (-DPT_EXP_DOUBLE_LOAD) and with the slab tests evaluated twice (-DPT_EXP_DOUBLE_VALU) and times the
# closest-hit leg of the bench with each (tools/try_variants.sh).  Build here, run on the GPU box:
#   bash tools/sensitivity.sh build && gpurun -- 'bash tools/sensitivity.sh run'
cd "$(dirname "$0")/.."
case "$1" in
  build)
    PTAMD_OUT=variant_dload.so bash cuda-path-tracer_amd/build.sh -DPT_EXP_DOUBLE_LOAD || exit 1
    PTAMD_OUT=variant_dvalu.so bash cuda-path-tracer_amd/build.sh -DPT_EXP_DOUBLE_VALU || exit 1
    ;;
  run)
    bash tools/try_variants.sh
    ;;
  *) echo "usage: $0 build|run"; exit 2;;
esac
