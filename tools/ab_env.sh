#!/bin/bash
# full-step A/B of an environment switch: tools/ab_env.sh NAME VALUE_A VALUE_B
for rep in 1 2 3; do
for v in $2 $3; do
  echo -n "$1=$v: "
  env $1=$v python bench.py --steps ${STEPS:-30} --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_ms'])"
done
done
