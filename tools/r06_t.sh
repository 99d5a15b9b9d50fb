#!/bin/bash
python tools/ab_attn_fwd.py libm3p_hip.so libm3p_hip_r05.so libm3p_hip_wl0.so libm3p_hip_fa1.so libm3p_hip_fa2.so libm3p_hip_fa3.so libm3p_hip.so 2>&1 | tee gpurun_out/r06/s6_attn_fwd_ab.txt
