for rep in 1 2; do
for lib in libm3p_hip.so libm3p_hip_alt8.so libm3p_hip_alt14.so libm3p_hip_alt16.so; do
  echo "== $lib"
  for cfg in "3072 768 0" "768 3072 0" "2304 768 1" "768 768 3"; do
    set -- $cfg
    M3P_HIP_LIB=$PWD/m3p_amd/$lib M3P_VARIANT=2 python tools/gemm_bench.py nt 41984 $1 $2 30 $3 2>&1 | tail -1
  done
done
done
