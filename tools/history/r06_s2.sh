#!/bin/bash
# round 6, session 2: cache policy of the output-row stores (0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt)
mkdir -p gpurun_out/r06
O=gpurun_out/r06
export AB_ONLY='QKV,FFN1 fwd,geluq,out_lin,FFN2,dx1,dh,dctx,mulq'
timeout 900 python tools/ab_gemm.py libm3p_hip.so:1 libm3p_hip_st1.so:1 libm3p_hip_st2.so:1 libm3p_hip_st3.so:1 libm3p_hip_st4.so:1 libm3p_hip.so:1 > $O/s2_store_policy.txt 2>&1
cat $O/s2_store_policy.txt
