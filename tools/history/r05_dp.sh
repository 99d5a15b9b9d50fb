#!/bin/bash
# round 5: one-rank wrapped step (every collective over RCCL on one GPU) against the unwrapped step, same box; and a two-rank dry
# run of bench.py's N > 1 half over gloo on the one GPU (the comm block with world = 2: bounds, split tail, ranks seen)
cd /root/repo
mkdir -p gpurun_out/r05
out=gpurun_out/r05/dp_one_rank_wrapped.txt
for i in 1 2; do
python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', d['ms_per_step'])" | tee -a $out
M3P_DP_FORCE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2961$i bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['comm']; print('wrapped zero1', d['ms_per_step'], c['exposed_ms_per_step'], c['exposed_split_ms'], c['rccl_ranks_seen'], {k:(v['MB'],v['ms'],v['lands']) for k,v in c['buckets'].items()})" | tee -a $out
done
M3P_DP_FORCE=1 M3P_DP_MODE=allreduce HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29618 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wrapped allreduce', d['ms_per_step'], d['comm']['exposed_ms_per_step'])" | tee -a $out
M3P_BENCH_SHARED_GPU=1 python bench.py --gpus 2 --steps 3 --warmup 1 --batch 32 --no-cpu-baseline > gpurun_out/r05/bench_two_ranks_shared_gpu.json 2> gpurun_out/r05/bench_two_ranks_shared_gpu.err
tail -1 gpurun_out/r05/bench_two_ranks_shared_gpu.json | cut -c1-200
