mkdir -p gpurun_out/r3m
rm -rf gpurun_out/counters
bash tools/collect_counters.sh > gpurun_out/r3m/collect.log 2>&1
python bench.py > gpurun_out/r3m/bench_default.json 2> gpurun_out/r3m/bench_default.err; tail -c 700 gpurun_out/r3m/bench_default.json
python bench.py --config cfg3 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r3m/bench_cfg3.json 2>/dev/null; cut -c1-260 gpurun_out/r3m/bench_cfg3.json
python bench.py --config cfg4 --batch 64 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r3m/bench_cfg4_bf16.json 2>/dev/null; cut -c1-260 gpurun_out/r3m/bench_cfg4_bf16.json
python bench.py --config cfg4 --batch 64 --fp8 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r3m/bench_cfg4_fp8.json 2>/dev/null; cut -c1-260 gpurun_out/r3m/bench_cfg4_fp8.json
python bench.py --ragged --no-cpu-baseline > gpurun_out/r3m/bench_ragged.json 2>/dev/null; cut -c1-200 gpurun_out/r3m/bench_ragged.json
