#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4n
python tools/ab_vocab_wgrad.py 2>&1 | tee gpurun_out/r4n/ab_vocab_wgrad.txt
timeout 900 python -m pytest tests/test_gemm.py tests/test_model_parity.py tests/test_full_size.py tests/test_small_kernels.py -m gpu -x -q > gpurun_out/r4n/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4n/pytest.log
tail -5 gpurun_out/r4n/pytest.log
tools/ab_bench.sh 2 > gpurun_out/r4n/ab_bench.txt 2>&1; cat gpurun_out/r4n/ab_bench.txt
