#!/bin/bash
# host runtime calls + kernels at the step boundary of the one-rank wrapped data-parallel step (run on the GPU box)
R=/root/repo
out=$R/gpurun_out/dpbound; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
M3P_DP_FORCE=1 M3P_DP_MODE=${DP_MODE:-zero1} HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29633 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $out/t -- python $R/bench.py --gpus 1 --steps 8 --warmup 8 --no-cpu-baseline > $out/run.log 2>&1
ls -la $out/t/*/
cd $R
python tools/dp_boundary.py $out/t --to seq_masks_kernel > $out/boundary.txt 2>&1
python tools/dp_boundary.py $out/t --frm gemm_nt_w4_kernel\<4 --to ln_bwd_hw_kernel --min-us 2 > $out/layer_bucket.txt 2>&1
rm -f $out/t/*/*trace.csv
head -60 $out/boundary.txt
