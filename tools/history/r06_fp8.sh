#!/bin/bash
# round 6: cfg4 (24L / 1024d, 100 + 256, batch 64) bf16 against --fp8 in the recipes of m3p_amd/fp8.py, alternating on one box;
# the fp8 tests first
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 1500 python -m pytest tests/test_fp8.py -x -q -m gpu > $O/fp8_tests.txt 2>&1; tail -4 $O/fp8_tests.txt
run() { python bench.py --config cfg4 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for r in 1 2; do
echo "bf16            $(run)"
echo "fp8 (default)   $(run --fp8)"
echo "fp8 r5 recipe   $(M3P_FP8_SITES=r5 run --fp8)"
echo "fp8 + qkv       $(M3P_FP8_SITES=qkv run --fp8)"
done | tee $O/fp8_cfg4_ab.txt
