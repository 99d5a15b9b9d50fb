#!/bin/bash
# round-4 collection: default bench line, counters + census, cfg3 / cfg4 lines, same-box A/B against the round-3 tree
cd /root/repo
mkdir -p gpurun_out/r4fin
python bench.py --steps 20 --warmup 5 > gpurun_out/r4fin/bench_default.json 2> gpurun_out/r4fin/bench_default.err; tail -1 gpurun_out/r4fin/bench_default.json | cut -c1-400
rm -rf gpurun_out/counters
timeout 1500 tools/collect_counters.sh > gpurun_out/r4fin/counters.log 2>&1
tail -3 gpurun_out/r4fin/counters.log
python bench.py --config cfg3 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4fin/bench_cfg3.json 2>/dev/null; tail -1 gpurun_out/r4fin/bench_cfg3.json | cut -c1-300
python bench.py --config cfg4 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4fin/bench_cfg4_bf16.json 2>/dev/null; tail -1 gpurun_out/r4fin/bench_cfg4_bf16.json | cut -c1-300
python bench.py --config cfg4 --batch 64 --fp8 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4fin/bench_cfg4_fp8.json 2>/dev/null; tail -1 gpurun_out/r4fin/bench_cfg4_fp8.json | cut -c1-300
python bench.py --ragged --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4fin/bench_ragged.json 2>/dev/null; tail -1 gpurun_out/r4fin/bench_ragged.json | cut -c1-300
tools/ab_bench.sh 2 > gpurun_out/r4fin/ab_bench.txt 2>&1; cat gpurun_out/r4fin/ab_bench.txt
