#!/bin/bash
# A/B of two library builds on NT GEMM shapes: tools/ab_libs.sh libA.so libB.so
for rep in 1 2; do
for lib in "$@"; do
  echo "== $lib"
  for cfg in "3072 768 1" "2304 768 1" "768 768 3" "768 3072 4" "768 2304 4" "3072 768 0"; do
    set -- $cfg
    M3P_HIP_LIB=$PWD/m3p_amd/$lib python tools/gemm_bench.py nt 41984 $1 $2 30 $3 2>&1 | tail -1
  done
done
done
