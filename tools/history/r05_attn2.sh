#!/bin/bash
mkdir -p gpurun_out/r05
# variants: 1 = two-phase (rounds 1-4), 2 = plain one-pass, 0 = persistent one-pass; bits 8.. = stagger (x 1024 clocks per phase)
timeout 600 python -m pytest tests/test_attention.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r05/attn_tests.log
timeout 600 python tools/ab_attn.py libm3p_hip.so:1 libm3p_hip.so:2 libm3p_hip.so:0 libm3p_hip.so:512 > gpurun_out/r05/attn_ab3.txt 2>&1
timeout 300 python tools/attn_stress.py 100 > gpurun_out/r05/attn_stress.txt 2>&1
for i in 1 2; do
  M3P_ATTN_VARIANT=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two-phase', d['ms_per_step'])"
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('persistent one-pass', d['ms_per_step'])"
done > gpurun_out/r05/ab_attn_step.txt 2>&1
