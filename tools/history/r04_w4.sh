#!/bin/bash
# new K-tile schedule of the four-wave NT kernel: correctness, A/B against the eight-wave kernel, the old schedule and the vendor, timeline
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/w4
if [ -z "$NO_TEST" ]; then timeout 900 python -m pytest tests/test_gemm.py -x -q -m gpu -k "256x256_kernels and 2-" 2>&1 | tail -5 | tee gpurun_out/w4/pytest.txt; fi
ARMS=${ARMS:-"libm3p_hip.so:1 libm3p_hip.so:2 libm3p_hip_w4old.so:2"}
AB_ONLY=${AB_ONLY:-"QKV,FFN1 fwd,out_lin,FFN2,dctx,dx1,dh"} timeout 600 python tools/ab_gemm.py $ARMS 2>&1 | tee gpurun_out/w4/ab.txt
python tools/vendor_names.py 2>&1 | tee gpurun_out/w4/vendor.txt
for abl in 0 512; do
  echo "== ablation $abl (256: no fragment reads, 512: no LDS-DMA)"
  M3P_VARIANT=$((1 + abl)) timeout 300 python tools/gemm_timeline.py 41984 768 3072
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/w4/timeline.txt
