for rep in 1 2; do
for lib in libm3p_hip.so libm3p_hip_b1024.so libm3p_hip_b2048.so; do
  echo -n "$lib: "
  M3P_HIP_LIB=$PWD/m3p_amd/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
done
