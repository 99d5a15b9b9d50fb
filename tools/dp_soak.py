#!/usr/bin/env python3
"""Soak of the wrapped step against the plain one: N optimizer steps of the benchmarked configuration (cfg2, dropout 0.1, fresh
synthetic batch every step, same seeds), first unwrapped, then as a forced one-rank world over RCCL in the given mode - the MLM
and ITM loss trajectories must coincide to the step's own noise (run-to-run fp32 atomics; dropout masks are counter-based and
identical).      python tools/dp_soak.py [steps] [zero1|allreduce]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench   # noqa: E402
from m3p_amd import synth   # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
mode = sys.argv[2] if len(sys.argv) > 2 else 'zero1'
cfg = dict(synth.CONFIGS['cfg2'])
cfg['B'] = 128


def run(wrapped):
    trainer, tup = bench.build(cfg, 0.1, 1, 0, 0, wrapped=wrapped, lr='0.0005')
    dev = torch.device('cuda', 0)
    out = []
    for s in range(steps):
        batch = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'], seed=5000 + s, ragged=False)
        img = batch['x_img'].transpose(0, 1).contiguous().to(dev)
        loc = batch['image_loc'].transpose(0, 1).contiguous().to(dev)
        t = ((batch['x'].to(dev), batch['lengths'].to(dev), batch['x_labels']),
             (img, torch.ones(cfg['B'], cfg['R'], dtype=torch.long, device=dev), loc, None, batch['pos_labels'].tolist(), None, None))
        trainer.pretrain_under_step(t, 'google', 't2i', 'en', 1.0, 1.0, 1.0, 1.0)
        trainer.n_iter += 1
    torch.cuda.synchronize()
    st = trainer.stats
    mlm = [float(v) for v in st['CMLM-google']]
    itm = [float(v) for v in st['t2i-google']]
    return mlm, itm


torch.cuda.set_device(0)
plain = run(False)
plain2 = run(False)
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29733', RANK='0', WORLD_SIZE='1', HSA_ENABLE_IPC_MODE_LEGACY='0',
                  M3P_DP_MODE=mode, M3P_DP_FORCE='1')
import torch.distributed as dist   # noqa: E402
dist.init_process_group('nccl', rank=0, world_size=1)
wrapped = run(True)
dist.destroy_process_group()
print('step   mlm plain / plain again / wrapped        itm plain / plain again / wrapped')
for s in list(range(0, steps, max(steps // 10, 1))) + [steps - 1]:
    print('%4d   %.4f / %.4f / %.4f        %.4f / %.4f / %.4f' % (s, plain[0][s], plain2[0][s], wrapped[0][s], plain[1][s], plain2[1][s], wrapped[1][s]))
k = max(steps // 6, 1)
avg = lambda v: sum(v[-k:]) / k   # noqa: E731
noise = abs(avg(plain[0]) - avg(plain2[0]))
diff = abs(avg(plain[0]) - avg(wrapped[0]))
print('mean MLM loss of the last %d steps: plain %.4f, plain again %.4f, wrapped (%s) %.4f   |wrapped - plain| %.4f, run-to-run %.4f'
      % (k, avg(plain[0]), avg(plain2[0]), mode, avg(wrapped[0]), diff, noise))
assert diff <= max(5 * noise, 0.02), 'the wrapped step drifts from the plain one'
print('ok')
