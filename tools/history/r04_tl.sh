#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/w4
for abl in 0 256 512 768; do
  echo "== ablation $abl (256: no fragment reads, 512: no LDS-DMA)"
  M3P_VARIANT=$((1 + abl)) timeout 300 python tools/gemm_timeline.py 41984 768 3072
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/w4/timeline.txt
