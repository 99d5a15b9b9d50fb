#!/bin/bash
mkdir -p gpurun_out/r05
timeout 600 python tools/ab_adam.py libm3p_hip.so libm3p_hip_adamq4.so libm3p_hip_adamq1.so libm3p_hip_adamb2k.so libm3p_hip_adamb8k.so libm3p_hip_adamb16k.so libm3p_hip_adamq4b8k.so > gpurun_out/r05/adam_ab.txt 2>&1
