#!/bin/bash
# idle GPU time of the one-rank wrapped data-parallel step (every collective over RCCL on one GPU), by (previous -> next kernel);
# the plain step beside it.   (round 4's r04_dp_gaps.sh / r04_plain_gaps.sh in one; run on the GPU box)
R=/root/repo
out=$R/gpurun_out/dpgaps; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
M3P_DP_FORCE=1 M3P_DP_MODE=${DP_MODE:-zero1} HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29633 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/wrapped -- python $R/bench.py --gpus 1 --steps 12 --warmup 5 --no-cpu-baseline > $out/wrapped.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/plain -- python $R/bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-also > $out/plain.log 2>&1
cd $R
tail -1 $out/wrapped.log | cut -c1-200; tail -1 $out/plain.log | cut -c1-200
t=$(ls $out/wrapped/*/*kernel_trace.csv | head -1)
python tools/trace_gaps.py $t --steps 3 --top 24 --marker ce_grad_tile_kernel --boundary adam_ranges_kernel,gemm_nt_w8_kernel 2>&1 | tee $out/gaps_wrapped.txt
p=$(ls $out/plain/*/*kernel_trace.csv | head -1)
python tools/trace_gaps.py $p --steps 3 --top 12 --marker ce_grad_tile_kernel 2>&1 | tee $out/gaps_plain.txt
python tools/trace_diff.py $t $p --steps 3 2>&1 | tee $out/diff.txt
rm -f $out/*/*/*trace.csv
