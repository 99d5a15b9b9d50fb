#!/bin/bash
# kernel census of the one-rank wrapped (zero1) step against the plain step: where the wrap's own time goes
R=$(cd "$(dirname "$0")/../.." && pwd)
out=$R/gpurun_out/dpprof; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/plain -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also > $out/plain.log 2>&1
M3P_DP_FORCE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29631 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/wrapped -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/wrapped.log 2>&1
tail -1 $out/plain.log | cut -c1-200; tail -1 $out/wrapped.log | cut -c1-200
cd $R
a=$(ls $out/plain/*/*kernel_stats.csv | head -1); b=$(ls $out/wrapped/*/*kernel_stats.csv | head -1)
cp $a $out/plain_kernel_stats.csv; cp $b $out/wrapped_kernel_stats.csv
python tools/kstats_diff.py $a $b 2>&1 | head -40 | tee $out/diff.txt
find $out -name "*kernel_trace.csv" -delete
