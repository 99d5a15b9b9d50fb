#!/bin/bash
for r in 1 2 3; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('draw in kernel', d['ms_per_step'])"
M3P_ATTN_KEEP_AHEAD=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('words ahead on a side stream', d['ms_per_step'])"
done
