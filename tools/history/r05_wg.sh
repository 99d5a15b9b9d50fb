#!/bin/bash
# round 5: the weight-gradient kernel with its in-launch ordered reduction - tests, stress loop, in-process A/B against the
# round-4 build (m3p_amd/libm3p_hip_base.so = _ab/base's library), whole-tile parity cases, step A/B
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_gemm.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r05/wg_tests.log
timeout 300 python tools/ab_wgrad.py libm3p_hip_base.so libm3p_hip.so --check > gpurun_out/r05/wg_ab.txt 2>&1
timeout 600 python tools/wgrad_stress2.py 600 > gpurun_out/r05/wg_stress.txt 2>&1
timeout 900 python -m pytest tests/test_model_parity.py -x -q -m gpu -k "tiles" 2>&1 | tail -15 > gpurun_out/r05/tiles_tests.log
NEWARGS=--no-also timeout 600 bash tools/ab_bench.sh 2 > gpurun_out/r05/ab_vs_r04_wg.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --instances > gpurun_out/r05/bench_instances.json 2>/dev/null
