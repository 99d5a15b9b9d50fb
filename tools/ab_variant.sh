#!/bin/bash
# A/B of GEMM kernel generations (M3P_VARIANT) on the cfg2 shapes, interleaved in one GPU session
for rep in 1 2; do
for v in ${VARIANTS:-1 2}; do
  for cfg in "3072 768 0" "768 3072 0" "2304 768 1" "768 768 0" "3072 768 5" "768 3072 3"; do
    set -- $cfg
    M3P_VARIANT=$v python tools/gemm_bench.py nt 41984 $1 $2 30 $3 2>&1 | tail -1
  done
done
done
