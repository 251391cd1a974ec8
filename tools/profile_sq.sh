#!/bin/bash
# Shader-engine counter passes over the closest-hit leg only (no render leg, no CPU leg).
set -e -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/sq
rm -rf $OUT && mkdir -p $OUT
SHORT="bench.py --steps 3 --warmup 1 --cpu-seconds 0 --spp-steps ${SQ_SPP_STEPS:-0}"
timeout -s KILL 60 rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || true
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -- python3 $SHORT > $OUT/a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_INSTS_SMEM --output-format csv -d $OUT/b -- python3 $SHORT > $OUT/b.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $OUT/c -- python3 $SHORT > $OUT/c.log 2>&1 || echo "pass c failed"
ls $OUT
