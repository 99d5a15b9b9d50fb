#!/bin/bash
python -m pytest tests/test_fp8.py -q -m gpu -k "8bit_copy" 2>&1 | tail -15
python -m pytest tests/test_gemm.py -q -m gpu -k "gelu_byte" 2>&1 | tail -3
