#!/bin/bash
# round 5: the whole GPU suite + the default bench line (with its cfg3 leg)
mkdir -p gpurun_out/r05
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/r05/full_gpu_tests.log
