#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4f
timeout 900 python -m pytest tests/test_gemm.py tests/test_small_kernels.py tests/test_model_parity.py tests/test_full_size.py -m gpu -x -q > gpurun_out/r4f/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4f/pytest.log
tail -12 gpurun_out/r4f/pytest.log
AB_ONLY="vocab" python tools/ab_gemm.py libm3p_hip.so:1 > gpurun_out/r4f/ab_gemm.txt 2>&1; cat gpurun_out/r4f/ab_gemm.txt
for v in 0 1 0 1; do M3P_CE_FUSED_LSE=$v python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused_lse=$v', d['ms_per_step'])"; done | tee gpurun_out/r4f/ab_step.txt
