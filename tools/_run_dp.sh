run() { env "$@" M3P_DP_FORCE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['roofline']['avg_ms'], d.get('comm',{}).get('exposed_ms_per_step'))"; }
python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('unwrapped', d['ms_per_step'], d['roofline']['avg_ms'])"
run M3P_DP_MODE=zero1
run M3P_DP_MODE=zero1 M3P_DP_TILE_QUEUE=0
run M3P_DP_MODE=zero1
run M3P_DP_MODE=zero1 M3P_DP_TILE_QUEUE=0
