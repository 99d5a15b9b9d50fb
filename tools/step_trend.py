import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from m3p_amd import synth
cfg = dict(synth.CONFIGS['cfg2'])
trainer, tup = bench.build(cfg, 0.1, 1, 0, 0)
ts = []
for i in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    trainer.pretrain_under_step(tup, 'google', 't2i', 'en', 1.0, 1.0, 1.0, 1.0); trainer.n_iter += 1
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(' '.join('%.1f' % t for t in ts))
