#!/bin/bash
# Same-box A/B of the benchmarked step: the tree under _ab/base (a `git archive` of an earlier commit, built separately)
# against the working tree, alternating, N rounds.   tools/ab_bench.sh [rounds] [extra bench.py args...]
#   -> gpurun_out/ab/{base,new}_<i>.json and a summary line per run; NEWARGS = arguments only the working tree's bench.py knows
R=${1:-2}; shift
mkdir -p gpurun_out/ab
for i in $(seq 1 $R); do
  (cd _ab/base && python bench.py --no-cpu-baseline --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1) > gpurun_out/ab/base_$i.json
  python bench.py --no-cpu-baseline --steps 30 --warmup 5 $NEWARGS "$@" 2>/dev/null | tail -1 > gpurun_out/ab/new_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['avg_ms'])
    except Exception as e: print(f, 'ERR', e)
PY
