#!/bin/bash
# round-5 collection: default bench line (with its cfg3 leg), a 100-step line, counters + census, cfg4 lines, ragged,
# same-box A/B against the round-4 tree (_ab/base), the in-step GEMM instance table
cd /root/repo
mkdir -p gpurun_out/r5fin
python bench.py --steps 20 --warmup 5 > gpurun_out/r5fin/bench_default.json 2> gpurun_out/r5fin/bench_default.err; tail -1 gpurun_out/r5fin/bench_default.json | cut -c1-300
python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-also --instances > gpurun_out/r5fin/bench_100_steps.json 2>/dev/null; tail -1 gpurun_out/r5fin/bench_100_steps.json | cut -c1-200
NEWARGS=--no-also tools/ab_bench.sh 2 > gpurun_out/r5fin/ab_bench.txt 2>&1; cat gpurun_out/r5fin/ab_bench.txt
rm -rf gpurun_out/counters
timeout 1500 tools/collect_counters.sh --no-also > gpurun_out/r5fin/counters.log 2>&1
tail -3 gpurun_out/r5fin/counters.log
python bench.py --config cfg3 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r5fin/bench_cfg3.json 2>/dev/null; tail -1 gpurun_out/r5fin/bench_cfg3.json | cut -c1-300
python bench.py --config cfg4 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r5fin/bench_cfg4_bf16.json 2>/dev/null; tail -1 gpurun_out/r5fin/bench_cfg4_bf16.json | cut -c1-300
python bench.py --ragged --steps 10 --warmup 3 --no-cpu-baseline --no-also > gpurun_out/r5fin/bench_ragged.json 2>/dev/null; tail -1 gpurun_out/r5fin/bench_ragged.json | cut -c1-300
