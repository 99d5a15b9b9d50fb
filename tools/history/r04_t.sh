#!/bin/bash
# one-rank wrapped step (every collective over RCCL on one GPU) against the unwrapped step, same box
cd /root/repo
mkdir -p gpurun_out/r4t
python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', d['ms_per_step'])" | tee gpurun_out/r4t/dp1.txt
for q in 1 0; do
M3P_DP_FORCE=1 M3P_DP_TILE_QUEUE=$q HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wrapped zero1 queue=$q', d['ms_per_step'], d['comm']['exposed_ms_per_step'], {k:(v['MB'],v['ms']) for k,v in d['comm']['buckets'].items()})" | tee -a gpurun_out/r4t/dp1.txt
done
M3P_DP_FORCE=1 M3P_DP_MODE=allreduce HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29618 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wrapped allreduce', d['ms_per_step'], d['comm']['exposed_ms_per_step'])" | tee -a gpurun_out/r4t/dp1.txt
