#!/bin/bash
# round 5, first GPU session: the new whole-tile model-level parity cases, the kernel tests the round's first edits touch,
# the default bench line (with its cfg3 leg) and a same-box A/B against the round-4 tree (_ab/base)
mkdir -p gpurun_out/r05
python -m pytest tests/test_model_parity.py -x -q -m gpu -k "tiles or cfg1" 2>&1 | tail -15 > gpurun_out/r05/tiles_tests.log
python -m pytest tests/test_gemm.py tests/test_small_kernels.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r05/gemm_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err
NEWARGS=--no-also bash tools/ab_bench.sh 2 > gpurun_out/r05/ab_vs_r04.txt 2>&1
