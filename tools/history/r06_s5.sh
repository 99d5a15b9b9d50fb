#!/bin/bash
# round 6, session 5: the 8-instruction dropout hash + batched keep-word writes - whole GPU suite, attention timing, step A/B
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 2400 python -m pytest tests -x -q -m gpu > $O/s5_gpu_tests.txt 2>&1; tail -4 $O/s5_gpu_tests.txt
python tools/attn_bench.py > $O/s5_attn_bench.txt 2>&1; tail -8 $O/s5_attn_bench.txt
NEWARGS=--no-also tools/ab_bench.sh 3 > $O/s5_ab_bench.txt 2>&1; cat $O/s5_ab_bench.txt
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --instances 2>/dev/null | tail -1 > $O/s5_bench_instances.json
