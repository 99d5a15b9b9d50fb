#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4r
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4r/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4r/pytest.log
tail -6 gpurun_out/r4r/pytest.log
cp _ab/base/m3p_amd/libm3p_hip.so m3p_amd/libm3p_hip_r4start.so
AB_ONLY="out_lin fwd,FFN2 fwd,dctx" python tools/ab_gemm.py libm3p_hip_r4start.so:1 libm3p_hip.so:1 > gpurun_out/r4r/ab_gemm.txt 2>&1; cat gpurun_out/r4r/ab_gemm.txt
python tools/ab_attn.py libm3p_hip_r4start.so libm3p_hip.so > gpurun_out/r4r/ab_attn.txt 2>&1; tail -5 gpurun_out/r4r/ab_attn.txt
tools/ab_bench.sh 2 > gpurun_out/r4r/ab_bench.txt 2>&1; cat gpurun_out/r4r/ab_bench.txt
