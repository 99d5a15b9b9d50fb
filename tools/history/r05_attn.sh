#!/bin/bash
# round 5: the persistent one-pass attention backward - parity tests, in-process A/B against the round-4 library, stress loop, step A/B
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_attention.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r05/attn_tests.log
timeout 300 python tools/ab_attn.py libm3p_hip_base.so libm3p_hip.so > gpurun_out/r05/attn_ab.txt 2>&1
timeout 300 python tools/attn_stress.py 120 > gpurun_out/r05/attn_stress.txt 2>&1
timeout 900 python -m pytest tests/test_model_parity.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r05/parity_tests.log
for i in 1 2; do
  M3P_ATTN_VARIANT=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two-phase', d['ms_per_step'])"
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('persistent one-pass', d['ms_per_step'])"
done > gpurun_out/r05/ab_attn_step.txt 2>&1
