#!/bin/bash
mkdir -p gpurun_out/r05
M3P_HIP_LIB=m3p_amd/libm3p_hip_alt.so timeout 300 python tools/attn_bwd_p_timeline.py > gpurun_out/r05/attn_bwd_persistent_timeline.txt 2>&1
