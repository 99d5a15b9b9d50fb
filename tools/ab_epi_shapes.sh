#!/bin/bash
# A/B of two library builds over the aux-reading epilogues on the cfg2 shapes: tools/ab_epi_shapes.sh libA.so libB.so
for lib in "$@" "$@"; do
  echo "== $lib"
  for cfg in "3072 768 5" "768 3072 4" "768 768 4" "768 768 3" "768 2304 4" "768 3072 3"; do
    set -- $cfg
    M3P_HIP_LIB=$PWD/m3p_amd/$lib python tools/gemm_bench.py nt 41984 $1 $2 30 $3 2>&1 | tail -1
  done
done
