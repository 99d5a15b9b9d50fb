#!/bin/bash
cd /root/repo
python tools/attn_bench.py 2>&1 | grep "p_drop=0.1"
python -m pytest tests/test_attention.py tests/test_refiner.py tests/test_model_parity.py tests/test_decoder.py -m gpu -x -q 2>&1 | tail -2
tools/ab_bench.sh 2 2>&1 | tail -4
