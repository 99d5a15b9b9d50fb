#!/bin/bash
# round-6 collection on one box: default bench line (with its cfg3 leg), a 100-step line with the in-step GEMM instance table,
# same-box A/B against the round-5 tree (_ab/base), counters + census at cfg2 and at cfg3, cfg4 bf16 / fp8, ragged
cd /root/repo
O=gpurun_out/r6fin; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.json | cut -c1-300
python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-also --instances 2>/dev/null | tail -1 > $O/bench_100_steps.json; cut -c1-200 $O/bench_100_steps.json
NEWARGS=--no-also tools/ab_bench.sh 3 > $O/ab_bench.txt 2>&1; cat $O/ab_bench.txt
rm -rf gpurun_out/counters gpurun_out/counters_cfg3
timeout 1500 tools/collect_counters.sh --no-also > $O/counters.log 2>&1; tail -2 $O/counters.log
COUNTERS_DIR=counters_cfg3 STATS_STEPS=12 timeout 2400 tools/collect_counters.sh --config cfg3 > $O/counters_cfg3.log 2>&1; tail -2 $O/counters_cfg3.log
python bench.py --config cfg3 --steps 10 --warmup 3 --no-cpu-baseline --instances 2>/dev/null | tail -1 > $O/bench_cfg3.json; cut -c1-300 $O/bench_cfg3.json
python bench.py --config cfg4 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_cfg4_bf16.json; cut -c1-200 $O/bench_cfg4_bf16.json
python bench.py --config cfg4 --batch 64 --fp8 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_cfg4_fp8.json; cut -c1-200 $O/bench_cfg4_fp8.json
python bench.py --ragged --steps 10 --warmup 3 --no-cpu-baseline --no-also 2>/dev/null | tail -1 > $O/bench_ragged.json; cut -c1-200 $O/bench_ragged.json
# keep only what gpurun can carry home (64 MiB): the per-dispatch traces are large
find gpurun_out/counters gpurun_out/counters_cfg3 -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out/counters gpurun_out/counters_cfg3
