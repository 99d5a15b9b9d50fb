#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out/vendor
python tools/vendor_names.py > gpurun_out/vendor/plain.txt 2>&1
AB_ONLY="FFN2,dx1,dh,FFN1 fwd,dctx,QKV" python tools/ab_gemm.py libm3p_hip.so:1 > gpurun_out/vendor/ours.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/vendor/prof -- python $R/tools/vendor_names.py > $R/gpurun_out/vendor/out.txt 2>&1
cd $R
find gpurun_out/vendor/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/vendor/kernel_stats.csv
rm -rf gpurun_out/vendor/prof
cat gpurun_out/vendor/plain.txt gpurun_out/vendor/ours.txt
cut -c1-600 gpurun_out/vendor/kernel_stats.csv | head -12
