#!/usr/bin/env python3
"""The FFN activation pass in its two forms at the benchmarked size: h = gelu(u) alone (round 3) against h + the one-byte
derivative in the dU GEMM's fragment order (round 4).  Alternating rounds, median; bytes moved and TB/s beside the times."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m3p_amd import ops   # noqa: E402

M, N = int(os.environ.get('AB_M', '41984')), 3072
u = torch.randn(M, N, device='cuda').to(torch.bfloat16)
arms = [('gelu_fwd', lambda: ops.gelu_fwd(u), 4 * M * N), ('gelu_fwd_gq', lambda: ops.gelu_fwd_gq(u), 5 * M * N)]
times = [[] for _ in arms]
for _, fn, _ in arms:
    for _ in range(3):
        fn()
for rnd in range(7):
    for i, (_, fn, _) in enumerate(arms):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times[i].append(e0.elapsed_time(e1) / 10)
for (name, _, nbytes), t in zip(arms, times):
    m = sorted(t)[len(t) // 2]
    print('%-14s %7.1f us   %6.1f MB   %5.2f TB/s' % (name, m * 1e3, nbytes / 1e6, nbytes / m / 1e9))
