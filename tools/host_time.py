#!/usr/bin/env python3
"""How long does the host need to enqueue one training step?  (If it is close to the GPU time per
step the GPU starves between kernels.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from m3p_amd import synth, ops
cfg = dict(synth.CONFIGS['cfg2']); cfg['B'] = 256
trainer, tup = bench.build(cfg, 0.1, 1, 0, 0)
def step():
    trainer.pretrain_under_step(tup, 'google', 't2i', 'en', 1.0, 1.0, 1.0, 1.0); trainer.n_iter += 1
for _ in range(4): step()
torch.cuda.synchronize()
for prof in (None, {}):
    ops.PROFILE = prof
    torch.cuda.synchronize()
    hs = []
    t0 = time.perf_counter()
    for _ in range(6):
        a = time.perf_counter(); step(); hs.append((time.perf_counter() - a) * 1e3)
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) * 1e3 / 6
    print('GEMM event hooks %s: host enqueue per step %s ms; wall per step %.2f ms' % ('on' if prof is not None else 'off', ['%.1f' % h for h in hs], tot), flush=True)
