#!/bin/bash
# A/B of two library builds on the full training step, interleaved in one GPU session
for rep in 1 2; do
for lib in m3p_amd/libm3p_hip.so m3p_amd/libm3p_hip_alt.so; do
  echo "== $lib"
  M3P_HIP_LIB=$PWD/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_ms'])"
done
done
