#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4o
timeout 600 python -m pytest tests/test_gemm.py -m gpu -x -q -k "gemm_nn or wgrad" 2>&1 | tail -4
python tools/ab_vocab_dgrad.py 2>&1 | tee gpurun_out/r4o/ab_vocab_dgrad.txt
timeout 900 python -m pytest tests/test_model_parity.py tests/test_full_size.py tests/test_small_kernels.py tests/test_kernel_isa.py -m gpu -x -q 2>&1 | tail -3
for v in 0 1 0 1; do M3P_VOCAB_DGRAD_W4=$v python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dgrad_w4=$v', d['ms_per_step'])"; done | tee gpurun_out/r4o/ab_step.txt
