#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4d
AB_ONLY="FFN1" python tools/ab_gemm.py libm3p_hip.so:1 libm3p_hip.so:6 > gpurun_out/r4d/ab_gemm.txt 2>&1; cat gpurun_out/r4d/ab_gemm.txt
