#!/bin/bash
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_model_parity.py -x -q -m gpu -k lazy 2>&1 | tail -30 > gpurun_out/r05/lazy_test.log
