#!/usr/bin/env python3
"""The vocabulary matrix's weight gradient (dE [250112, 768] from 4864 rows) accumulated with atomics against stored."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m3p_amd import ops   # noqa: E402
M, N, K = 4864, 250112, 768
dy = (torch.randn(M, N, device='cuda') * 0.01).to(torch.bfloat16)
x = torch.randn(M, K, device='cuda').to(torch.bfloat16)
dw = torch.zeros(N, K, device='cuda')
for name, kw in (('accumulate (atomics)', {}), ('store', dict(dw_is_zero=True))):
    for _ in range(3):
        ops.gemm_wgrad(dy, x, dw, **kw)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.gemm_wgrad(dy, x, dw, **kw)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    m = sorted(ts)[2]
    print('%-22s %8.1f us  %5.0f TF' % (name, m * 1e3, 2.0 * M * N * K / m / 1e9))
