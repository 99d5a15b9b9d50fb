#!/bin/bash
# Per-kernel HBM traffic and MFMA utilisation of one training step, for profiles/.
# Separate rocprofv3 --pmc passes (kernel-trace only), as MI355X_MICROARCH.md prescribes:
#   pass 1: FETCH_SIZE        pass 2: WRITE_SIZE        pass 3: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
# plus one plain --kernel-trace --stats pass for undisturbed durations.  Output: gpurun_out/counters/
# usage (on the GPU box): tools/collect_counters.sh
set -u
out=/root/repo/gpurun_out/counters
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
cmd="python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $cmd > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- $cmd > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- $cmd > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $out/sq -- $cmd > /dev/null 2>&1
ls $out/*/*/ | head -20
