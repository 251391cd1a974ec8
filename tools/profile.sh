#!/bin/bash
# Profiling passes of the headline bench on the GPU box.  Every pass is its own bounded run;
# the script stops at the first failure.  Results land under gpurun_out/prof/.
set -e -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
hipcc --offload-arch=gfx950 -O2 -o /tmp/pmc_calib tools/ubench/pmc_calib.hip 2>/dev/null
BENCH="bench.py --steps 20 --warmup 3 --cpu-seconds 0"
SHORT="bench.py --steps 3 --warmup 1 --cpu-seconds 0 --spp-steps 8 --spp-batch 8"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $BENCH > $OUT/bench_kt.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $SHORT > $OUT/bench_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $SHORT > $OUT/bench_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -- python3 $SHORT > $OUT/bench_l2.log 2>&1
timeout -s KILL 60 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- /tmp/pmc_calib > $OUT/calib_fetch.log 2>&1
timeout -s KILL 60 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- /tmp/pmc_calib > $OUT/calib_write.log 2>&1
timeout -k 10 300 python3 -u $BENCH > $OUT/bench_plain.log 2>&1
tail -1 $OUT/bench_plain.log
find $OUT -name "*.csv" | head -40
