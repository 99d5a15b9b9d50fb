#!/bin/bash
# round 6, session 1: where the byte-epilogue GEMMs' time goes (ablation builds, strip widths, half the chip, timeline)
mkdir -p gpurun_out/r06
O=gpurun_out/r06
export AB_ONLY='FFN1 fwd,geluq,mulq'
S=$((1<<16))
timeout 600 python -m pytest tests/test_gemm.py -x -q -m gpu -k "gelu_byte" > $O/s1_test.txt 2>&1; tail -3 $O/s1_test.txt
echo "== ablations (GQ_ABL: 1 no math, 2 no code store, 4 no h store; MQ_ABL: 1 no code read, 2 no dU store, 4 no colsum; a7 = GQ 7 / MQ 3; a6 = GQ 6 / MQ 7; pf = MQ prefetch, GQ 3)" > $O/s1_abl.txt
timeout 900 python tools/ab_gemm.py libm3p_hip.so:1 libm3p_hip_a1.so:1 libm3p_hip_a2.so:1 libm3p_hip_a4.so:1 libm3p_hip_a6.so:1 libm3p_hip_a7.so:1 libm3p_hip_pf.so:1 libm3p_hip.so:1 >> $O/s1_abl.txt 2>&1
cat $O/s1_abl.txt
echo "== strip widths (variant = 1 + (w + 16 serpentine) << 16): default(4), 12, 6, 3, 2, 6s, 4s" > $O/s1_strip.txt
timeout 900 python tools/ab_gemm.py libm3p_hip.so:1 libm3p_hip.so:$((1+12*S)) libm3p_hip.so:$((1+6*S)) libm3p_hip.so:$((1+3*S)) libm3p_hip.so:$((1+2*S)) libm3p_hip.so:$((1+22*S)) libm3p_hip.so:$((1+20*S)) >> $O/s1_strip.txt 2>&1
cat $O/s1_strip.txt
echo "== half the chip: 128 workgroups, M = 20992 (same tiles per workgroup)" > $O/s1_half.txt
AB_GRID=128 AB_M=20992 timeout 900 python tools/ab_gemm.py libm3p_hip.so:1 libm3p_hip_a6.so:1 libm3p_hip_a7.so:1 libm3p_hip_pf.so:1 >> $O/s1_half.txt 2>&1
echo "== quarter: 64 workgroups, M = 10496" >> $O/s1_half.txt
AB_GRID=64 AB_M=10496 timeout 900 python tools/ab_gemm.py libm3p_hip.so:1 libm3p_hip_a6.so:1 libm3p_hip_a7.so:1 >> $O/s1_half.txt 2>&1
cat $O/s1_half.txt
M3P_HIP_LIB=m3p_amd/libm3p_hip_tl.so timeout 600 python tools/w8_timeline.py > $O/s1_timeline.txt 2>&1
cat $O/s1_timeline.txt
