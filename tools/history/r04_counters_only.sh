#!/bin/bash
cd /root/repo
rm -rf gpurun_out/counters
mkdir -p gpurun_out/r4z; timeout 1500 tools/collect_counters.sh > gpurun_out/r4z/counters.log 2>&1
tail -3 gpurun_out/r4z/counters.log
