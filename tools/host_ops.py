#!/usr/bin/env python3
"""Which aten ops (fills, copies, casts, index ops ...) does one training step launch outside the C-ABI library, and
from where?  torch.profiler over two steps, grouped by op and Python call site."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from m3p_amd import synth
cfg = dict(synth.CONFIGS['cfg2']); cfg['B'] = 256
trainer, tup = bench.build(cfg, 0.1, 1, 0, 0)
def step():
    trainer.pretrain_under_step(tup, 'google', 't2i', 'en', 1.0, 1.0, 1.0, 1.0); trainer.n_iter += 1
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(2): step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=6).table(sort_by='device_time_total', row_limit=60, max_src_column_width=110, max_name_column_width=50))
