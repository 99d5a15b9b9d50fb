#!/bin/bash
# round 5: the whole GPU suite, then the final collection
mkdir -p gpurun_out/r05
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/r05/full_gpu_tests.log
bash tools/r05_final.sh > gpurun_out/r05/final.log 2>&1
