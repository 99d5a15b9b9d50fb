#!/bin/bash
# kernel statistics of the cfg3 line (1024 sequences per GPU, the batch the 0.40 target is stated on)
cd /tmp; export TMPDIR=/tmp
mkdir -p /root/repo/gpurun_out/r4c3
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r4c3/stats -- python /root/repo/bench.py --config cfg3 --steps 12 --warmup 3 --no-cpu-baseline > /root/repo/gpurun_out/r4c3/bench.log 2>&1
tail -1 /root/repo/gpurun_out/r4c3/bench.log | cut -c1-300
ls /root/repo/gpurun_out/r4c3/stats/*/
