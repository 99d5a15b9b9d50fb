#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4c
AB_ONLY="dU,FFN1" python tools/ab_gemm.py libm3p_hip.so:1 libm3p_hip_q1.so:1 libm3p_hip_q2.so:1 > gpurun_out/r4c/ab_gemm.txt 2>&1; cat gpurun_out/r4c/ab_gemm.txt
python tools/ab_gelu.py > gpurun_out/r4c/ab_gelu.txt 2>&1; cat gpurun_out/r4c/ab_gelu.txt
for l in libm3p_hip.so libm3p_hip_q1.so libm3p_hip_q2.so; do
  M3P_HIP_LIB=m3p_amd/$l python -m pytest tests/test_gemm.py -m gpu -x -q -k byte_derivative 2>&1 | tail -2
done
