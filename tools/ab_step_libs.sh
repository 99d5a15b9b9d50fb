for i in 1 2; do
for lib in m3p_amd/libm3p_hip_alt.so m3p_amd/libm3p_hip.so; do
  M3P_HIP_LIB=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['roofline']['avg_ms'], d['roofline']['frac'])"
done; done
