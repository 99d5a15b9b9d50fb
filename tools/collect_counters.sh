#!/bin/bash
# Per-kernel HBM traffic, MFMA utilisation, sustained clock and L2 hit rate of the training step, for profiles/.
# Separate rocprofv3 --pmc passes (kernel-trace only), as MI355X_MICROARCH.md prescribes:
#   fetch: FETCH_SIZE     write: WRITE_SIZE     sq: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
#   clk: GRBM_GUI_ACTIVE (sustained clock = counter / the dispatch's own duration in that pass)     tcc: TCC_HIT_sum TCC_MISS_sum
# plus one plain --kernel-trace --stats pass over 40 steady-state steps for undisturbed durations and the launch census.
# Output: gpurun_out/counters/ (COUNTERS_DIR=<name> for another directory; STATS_STEPS = steps of the plain pass)
# usage (on the GPU box): tools/collect_counters.sh [extra bench.py args]
set -u
out=/root/repo/gpurun_out/${COUNTERS_DIR:-counters}
STEPS=${STATS_STEPS:-40}
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
cmd="python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python /root/repo/bench.py --steps $STEPS --warmup 5 --no-cpu-baseline "$@" > $out/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- $cmd > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- $cmd > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $out/sq -- $cmd > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/clk -- $cmd > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $out/tcc -- $cmd > /dev/null 2>&1
ls $out/*/*/ | head -30
head -3 $out/clk/*/*counter_collection.csv
