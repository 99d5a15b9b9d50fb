#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4b
timeout 600 python -m pytest tests/test_gemm.py tests/test_lib_abi.py tests/test_model_parity.py tests/test_full_size.py -m gpu -x -q > gpurun_out/r4b/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4b/pytest.log
tail -15 gpurun_out/r4b/pytest.log
cp _ab/base/m3p_amd/libm3p_hip.so m3p_amd/libm3p_hip_r4start.so
python tools/ab_gemm.py libm3p_hip_r4start.so:1 libm3p_hip.so:1 > gpurun_out/r4b/ab_gemm.txt 2>&1; cat gpurun_out/r4b/ab_gemm.txt
tools/ab_bench.sh 2 > gpurun_out/r4b/ab_bench.txt 2>&1; cat gpurun_out/r4b/ab_bench.txt
M3P_GELU_BYTE_GRAD=0 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 | cut -c1-300
