#!/bin/bash
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_hw_probes.py tests/test_attention.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r05/attn_tests.log
cp _ab/base/m3p_amd/libm3p_hip.so m3p_amd/libm3p_hip_base.so 2>/dev/null
timeout 600 python tools/ab_attn.py libm3p_hip.so:1 libm3p_hip.so:0 libm3p_hip_prev.so:0 > gpurun_out/r05/attn_ab4.txt 2>&1
timeout 300 python tools/attn_stress.py 60 > gpurun_out/r05/attn_stress.txt 2>&1
