#!/usr/bin/env python3
"""Read-before-write hunt: fill the caching allocator's free blocks with 0xFF bytes (NaN in bf16 / fp32 / fp8) before a short
training run, so that any kernel reading a buffer nobody wrote turns the losses into NaN instead of into plausible numbers.
usage: python tools/poison_check.py [fp8|bf16] [small|cfg4x6]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import test_fp8 as T
from m3p_amd import synth

mode = sys.argv[1] if len(sys.argv) > 1 else 'fp8'
which = sys.argv[2] if len(sys.argv) > 2 else 'cfg4x6'
cfg = dict(emb_dim=256, n_heads=4, n_layers=4, n_words=8192, T=48, R=16, B=16, n_pred=8)
steps = 12
if which == 'cfg4x6':
    cfg = dict(synth.CONFIGS['cfg4']); cfg['n_layers'] = 6
if which == 'cfg2':
    cfg = dict(synth.CONFIGS['cfg2']); cfg['n_layers'] = 3; cfg['B'] = 64


def poison(gb):
    blocks = []
    for size in (1 << 30, 1 << 26, 1 << 22, 1 << 18):       # large and small pools, several block sizes
        n = max(1, int(gb * (1 << 30) / 4 / size))
        blocks += [torch.full((size,), 0xFF, dtype=torch.uint8, device='cuda') for _ in range(n)]
    torch.cuda.synchronize()
    del blocks


for rnd in range(2):
    poison(float(os.environ.get('POISON_GB', '48')))
    _, mlm, itm = T._train_curve(cfg, mode == 'fp8', steps)
    print(mode, which, 'round', rnd, 'mlm', np.round(mlm[:4], 4), '...', np.round(mlm[-2:], 4), 'itm', np.round(itm[:3], 4),
          'finite', bool(np.isfinite(mlm).all() and np.isfinite(itm).all()), flush=True)
