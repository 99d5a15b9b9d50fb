#!/bin/bash
# round 5: the whole GPU suite (+ the forward attention kernel's phase timeline on the instrumented build)
mkdir -p gpurun_out/r05
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/r05/full_gpu_tests.log
M3P_HIP_LIB=m3p_amd/libm3p_hip_alt.so timeout 300 python tools/attn_timeline.py > gpurun_out/r05/attn_fwd_timeline.txt 2>&1
