#!/bin/bash
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_attention.py tests/test_layernorm.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r05/attn_ln_tests.log
timeout 300 python tools/ab_ln.py libm3p_hip_lnold.so libm3p_hip.so > gpurun_out/r05/ln_ab.txt 2>&1
timeout 300 python tools/attn_bench.py > gpurun_out/r05/attn_bench.txt 2>&1
for i in 1 2; do
  M3P_HIP_LIB=m3p_amd/libm3p_hip_prev.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev (persistent bwd before the store / keep-word changes)', d['ms_per_step'])"
  M3P_HIP_LIB=m3p_amd/libm3p_hip_lnold.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new attention, old LayerNorm forward', d['ms_per_step'])"
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['ms_per_step'])"
done > gpurun_out/r05/ab_step_vmem.txt 2>&1
