#!/bin/bash
# full GPU suite on the working tree, then same-box A/B of the step: the committed build (libm3p_hip_base.so) against the working tree's
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/step
[ -n "$NO_TEST" ] || timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/step/pytest.txt
for i in 1 2 3; do
for lib in ${ABLIBS:-libm3p_hip_head.so libm3p_hip.so}; do
  M3P_HIP_LIB=$PWD/m3p_amd/$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['ms_per_step'], d['value'], d['roofline']['avg_ms'], d['roofline']['frac'])"
done; done | tee gpurun_out/step/ab.txt
