#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4i
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r4i/bench.json 2> gpurun_out/r4i/bench.err; tail -1 gpurun_out/r4i/bench.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r4i/stats -- python /root/repo/bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
ls /root/repo/gpurun_out/r4i/stats/*/ | head
