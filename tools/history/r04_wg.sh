#!/bin/bash
# buffer-form LDS-DMA in the weight-gradient kernel: correctness, then A/B against the committed build
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/wg
timeout 900 python -m pytest tests/test_gemm.py -x -q -m gpu -k "wgrad or gemm_nn" 2>&1 | tail -5 | tee gpurun_out/wg/pytest.txt
timeout 600 python tools/ab_wgrad.py libm3p_hip_base.so libm3p_hip.so --check 2>&1 | tee gpurun_out/wg/ab.txt
for l in libm3p_hip_base.so libm3p_hip.so libm3p_hip_base.so libm3p_hip.so; do
  echo "== $l"; M3P_HIP_LIB=$PWD/m3p_amd/$l timeout 600 python tools/ab_vocab_dgrad.py 2>&1 | tail -4; M3P_HIP_LIB=$PWD/m3p_amd/$l timeout 600 python tools/ab_vocab_wgrad.py 2>&1 | tail -3
done | grep -v amdgpu.ids | tee gpurun_out/wg/ab_vocab.txt
