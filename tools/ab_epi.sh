#!/bin/bash
# A/B of two library builds on the epilogue GEMM shapes, interleaved in one GPU session
for rep in 1 2; do
for lib in m3p_amd/libm3p_hip.so m3p_amd/libm3p_hip_alt.so; do
  echo "== $lib"
  for cfg in "3072 768 3" "3072 768 4" "3072 768 5" "768 3072 3" "768 3072 4" "768 768 3" "768 2304 4" "3072 768 0"; do
    set -- $cfg
    M3P_HIP_LIB=$PWD/$lib python tools/gemm_bench.py nt 41984 $1 $2 30 $3 2>&1 | tail -1
  done
done
done
