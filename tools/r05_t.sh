#!/bin/bash
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_small_kernels.py tests/test_attention.py tests/test_model_parity.py tests/test_distributed_gpu.py -q -m gpu 2>&1 | tail -12 > gpurun_out/r05/adam_attn_tests.log
