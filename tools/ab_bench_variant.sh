#!/bin/bash
# full-step A/B of GEMM kernel selections (M3P_VARIANT), interleaved in one GPU session
for rep in 1 2 3; do
for v in ${VARIANTS:-1 3}; do
  echo -n "variant $v: "
  M3P_VARIANT=$v python bench.py --steps ${STEPS:-30} --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_ms'])"
done
done
