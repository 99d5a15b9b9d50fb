#!/bin/bash
mkdir -p gpurun_out/r05
timeout 300 python tools/ab_wgrad.py libm3p_hip_base.so libm3p_hip.so --check > gpurun_out/r05/wg_ab2.txt 2>&1
timeout 600 python -m pytest tests/test_gemm.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r05/wg_tests2.log
timeout 300 python tools/wgrad_stress2.py 200 > gpurun_out/r05/wg_stress2.txt 2>&1
timeout 900 python -m pytest tests/test_model_parity.py -x -q -m gpu -k "tiles or lazy" 2>&1 | tail -15 > gpurun_out/r05/tiles_tests2.log
for i in 1 2; do
  M3P_LAZY_VOCAB_ZERO=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lazy0', d['ms_per_step'])"
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lazy1', d['ms_per_step'])"
  (cd _ab/base && python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base', d['ms_per_step'])")
done > gpurun_out/r05/ab_lazy.txt 2>&1
