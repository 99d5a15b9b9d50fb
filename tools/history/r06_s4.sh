#!/bin/bash
# round 6, session 4: step-level A/B against the round-5 tree (_ab/base), cfg3 line, in-step GEMM instance table
mkdir -p gpurun_out/r06
O=gpurun_out/r06
NEWARGS=--no-also tools/ab_bench.sh 3 > $O/s4_ab_bench.txt 2>&1; cat $O/s4_ab_bench.txt
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --instances 2>/dev/null | tail -1 > $O/s4_bench_instances.json; cut -c1-400 $O/s4_bench_instances.json
python bench.py --config cfg3 --steps 10 --warmup 3 --no-cpu-baseline --instances 2>/dev/null | tail -1 > $O/s4_bench_cfg3.json; cut -c1-400 $O/s4_bench_cfg3.json
