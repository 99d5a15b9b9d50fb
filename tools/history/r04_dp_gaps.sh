#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd)
out=$R/gpurun_out/dpgaps; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
M3P_DP_FORCE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29633 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/wrapped -- python $R/bench.py --gpus 1 --steps 12 --warmup 5 --no-cpu-baseline > $out/wrapped.log 2>&1
cd $R
t=$(ls $out/wrapped/*/*kernel_trace.csv | head -1)
python tools/trace_gaps.py $t --steps 3 --top 30 --marker ce_grad_tile_kernel 2>&1 | tee $out/gaps.txt
m=$(ls $out/wrapped/*/*memory_copy_trace.csv | head -1)
python tools/trace_window.py $t $m --from scatter_add_token_rows --to seq_masks_kernel 2>&1 | tee $out/window.txt
rm -f $out/wrapped/*/*trace.csv
