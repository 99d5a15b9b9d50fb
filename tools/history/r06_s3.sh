#!/bin/bash
# round 6, session 3: table-lookup GELU + byte epilogue - parity, A/B against the round-5 form, timeline
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 900 python -m pytest tests/test_gemm.py -x -q -m gpu > $O/s3_test_gemm.txt 2>&1; tail -5 $O/s3_test_gemm.txt
export AB_ONLY='FFN1 fwd,geluq,mulq'
echo "== main (LUT, nt stores, csum of fp32) | LUT off | nt off | csum of rounded | nt code loads | main" > $O/s3_ab.txt
timeout 900 python tools/ab_gemm.py libm3p_hip.so:1 libm3p_hip_lut0.so:1 libm3p_hip_gqnt0.so:1 libm3p_hip_mqc0.so:1 libm3p_hip_mqnl.so:1 libm3p_hip.so:1 >> $O/s3_ab.txt 2>&1
cat $O/s3_ab.txt
TL_ONLY='GELU,byte' M3P_HIP_LIB=m3p_amd/libm3p_hip_tl.so timeout 600 python tools/w8_timeline.py > $O/s3_timeline.txt 2>&1
cat $O/s3_timeline.txt
timeout 1500 python -m pytest tests/test_model_parity.py -x -q -m gpu > $O/s3_test_model.txt 2>&1; tail -5 $O/s3_test_model.txt
