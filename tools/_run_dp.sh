mkdir -p gpurun_out/r3c
python -m pytest tests -m gpu -x -q > gpurun_out/r3c/pytest.txt 2>&1; tail -4 gpurun_out/r3c/pytest.txt
python tools/cu_reserve_ab.py > gpurun_out/r3c/cu_reserve.txt 2> gpurun_out/r3c/cu_reserve.err; cat gpurun_out/r3c/cu_reserve.txt
cd /tmp; export TMPDIR=/tmp
cmd="python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /root/repo/gpurun_out/r3c/pmc_lds -- $cmd > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d /root/repo/gpurun_out/r3c/pmc_valu -- $cmd > /dev/null 2>&1
ls /root/repo/gpurun_out/r3c/pmc_lds/*/ | head
