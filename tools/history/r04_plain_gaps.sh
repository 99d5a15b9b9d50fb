#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd)
out=$R/gpurun_out/plaingaps; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/plain -- python $R/bench.py --steps 12 --warmup 5 --no-cpu-baseline > $out/plain.log 2>&1
cd $R
t=$(ls $out/plain/*/*kernel_trace.csv | head -1); m=$(ls $out/plain/*/*memory_copy_trace.csv | head -1)
python tools/trace_gaps.py $t --steps 3 --top 12 --marker ce_grad_tile_kernel 2>&1 | tee $out/gaps.txt
python tools/trace_window.py $t $m 2>&1 | tee $out/window.txt
rm -f $out/plain/*/*trace.csv
