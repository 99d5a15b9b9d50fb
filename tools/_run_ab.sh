mkdir -p gpurun_out/r3g
M3P_TILE_QUEUE=1 python -m pytest tests/test_gemm.py tests/test_model_parity.py -m gpu -x -q 2>&1 | tail -3
python -m pytest tests/test_gemm.py -m gpu -x -q 2>&1 | tail -2
python tools/cu_reserve_ab.py --queue-ab --launches 56 2>/dev/null | tee gpurun_out/r3g/queue_ab.txt
