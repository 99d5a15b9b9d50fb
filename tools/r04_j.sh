#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4j
AB_ONLY="out_lin,FFN2,dx1,dh,dctx,dU mulq,dU mul" python tools/ab_gemm.py libm3p_hip_q1.so:1 libm3p_hip_noaux.so:1 > gpurun_out/r4j/ab_gemm.txt 2>&1; cat gpurun_out/r4j/ab_gemm.txt
