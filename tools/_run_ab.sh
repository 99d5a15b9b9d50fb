mkdir -p gpurun_out/r3d
python tools/ab_attn.py libm3p_hip.so libm3p_hip_attn14.so libm3p_hip_attn26.so 2>&1 | grep -v Warn | tee gpurun_out/r3d/ab_attn.txt
python tools/ab_ln.py libm3p_hip.so libm3p_hip_lnbase.so libm3p_hip_lnb1024.so 2>&1 | grep -v Warn | tee gpurun_out/r3d/ab_ln.txt
python tools/ab_gemm.py libm3p_hip.so:1 libm3p_hip_sp2.so:1 libm3p_hip_sp0.so:1 2>&1 | grep -v Warn | tee gpurun_out/r3d/ab_gemm_setprio.txt
python -m pytest tests/test_attention.py tests/test_layernorm.py tests/test_distributed_gpu.py tests/test_streams_and_retrieval.py tests/test_model_parity.py -m gpu -x -q 2>&1 | tail -4
