#!/bin/bash
# round-4 opening run on the GPU box: the GPU suite, the default bench line, the counters + census
cd /root/repo
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4a/pytest.log
tail -5 gpurun_out/r4a/pytest.log
python bench.py --steps 30 --warmup 5 > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err; tail -1 gpurun_out/r4a/bench.json
rm -rf gpurun_out/counters
timeout 1200 tools/collect_counters.sh > gpurun_out/r4a/counters.log 2>&1
tail -8 gpurun_out/r4a/counters.log
