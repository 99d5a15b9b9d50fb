#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4k
M3P_HIP_LIB=m3p_amd/libm3p_hip_q3.so python -m pytest tests/test_gemm.py -m gpu -x -q -k byte_derivative 2>&1 | tail -3
AB_ONLY="dU mulq,FFN1 fwd" python tools/ab_gemm.py libm3p_hip.so:1 libm3p_hip_q1.so:1 libm3p_hip_q3.so:1 libm3p_hip_noaux.so:1 > gpurun_out/r4k/ab_gemm.txt 2>&1; cat gpurun_out/r4k/ab_gemm.txt
