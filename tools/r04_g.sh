#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4g
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r4g/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4g/pytest.log
tail -6 gpurun_out/r4g/pytest.log
