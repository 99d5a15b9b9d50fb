#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4e
AB_ONLY="FFN1,dU" python tools/ab_gemm.py libm3p_hip.so:1 > gpurun_out/r4e/ab_gemm.txt 2>&1; cat gpurun_out/r4e/ab_gemm.txt
timeout 900 python -m pytest tests/test_gemm.py tests/test_model_parity.py tests/test_full_size.py tests/test_distributed_gpu.py tests/test_streams_and_retrieval.py -m gpu -x -q > gpurun_out/r4e/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4e/pytest.log
tail -5 gpurun_out/r4e/pytest.log
tools/ab_bench.sh 2 > gpurun_out/r4e/ab_bench.txt 2>&1; cat gpurun_out/r4e/ab_bench.txt
