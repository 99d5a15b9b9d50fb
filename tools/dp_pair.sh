#!/bin/bash
# plain vs one-rank wrapped (every collective over RCCL on one GPU) training step, no profiler, interleaved repeats:
# what the data-parallel wrapper costs a step, and how far ahead of the GPU the host stays (host = time to ENQUEUE a step).
# legs: plain | single (wrapper present, world of one: no collective, no event) | allreduce | zero1   (LEGS="plain zero1" to choose)
R=/root/repo
out=$R/gpurun_out/dppair; mkdir -p $out
py='import json,sys
for l in sys.stdin:
    if l.startswith("{\"metric"):
        d = json.loads(l); print("%-10s %.3f ms/step   host enqueue %.2f ms/step" % (sys.argv[1], d["ms_per_step"], d["config"].get("host_enqueue_ms_per_step", 0)))'
port=29650
for i in 1 2 ${REPEATS:-}; do
  for leg in ${LEGS:-plain single allreduce zero1}; do
    port=$((port + 1))
    env="HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$port RANK=0 LOCAL_RANK=0 WORLD_SIZE=1"
    case $leg in
      plain)     python $R/bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-also 2>/dev/null | python -c "$py" plain ;;
      single)    env $env python $R/bench.py --gpus 1 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "$py" single ;;
      *)         env $env M3P_DP_FORCE=1 M3P_DP_MODE=$leg python $R/bench.py --gpus 1 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "$py" $leg ;;
    esac
  done
done 2>&1 | tee $out/pair.txt
