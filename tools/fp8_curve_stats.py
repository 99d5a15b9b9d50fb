#!/usr/bin/env python3
"""Spread of the statistics tests/test_fp8.py asserts on (fp8 against bf16 loss curves), over repeated runs - the step's
atomics make every run slightly different.  usage: python tools/fp8_curve_stats.py [repeats]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import test_fp8 as T
from m3p_amd import synth

run = lambda v: np.convolve(v, np.ones(4) / 4, mode='valid')      # noqa: E731
small = dict(emb_dim=256, n_heads=4, n_layers=4, n_words=8192, T=48, R=16, B=16, n_pred=8)
big = dict(synth.CONFIGS['cfg4']); big['n_layers'] = 6
LR = os.environ.get('CURVE_LR', '0.001,warmup_updates=8')
run8 = lambda v: np.convolve(v, np.ones(8) / 8, mode='valid')      # noqa: E731
for name, cfg, steps in (('small', small, 40), ('cfg4x6', big, 24)):
    for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
        _, m16, i16 = T._train_curve(cfg, False, steps, lr=LR)
        _, m8, i8 = T._train_curve(cfg, True, steps, lr=LR)
        def stats(ma, ia, mb, ib):
            return ((np.abs(run(ma) - run(mb)) / run(mb)).max(), abs(ma.mean() - mb.mean()) / mb.mean(),
                    (np.abs(run(ma + ia) - run(mb + ib)) / run(mb + ib)).max(), abs(ia.mean() - ib.mean()) / ib.mean(),
                    (np.abs(run8(ma + ia) - run8(mb + ib)) / run8(mb + ib)).max())
        print(name, r, 'fp8-bf16: mlm run4 max %.4f mean %.4f | total run4 max %.4f | itm mean %.4f | total run8 max %.4f' % stats(m8, i8, m16, i16), flush=True)
