#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4u
timeout 1200 python -m pytest tests/test_distributed_gpu.py -m gpu -x -q > gpurun_out/r4u/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4u/pytest.log
tail -6 gpurun_out/r4u/pytest.log
