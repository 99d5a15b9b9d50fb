#!/usr/bin/env python3
"""The vocabulary data gradient dH [4864, 768] += dlogits [4864, 250112] x E [250112, 768]: stream-K with atomics (round 3)
against the four-wave (tile, K-chunk) kernel with its first operand staged as an NT panel (round 4)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m3p_amd import ops   # noqa: E402
M, N, K = 4864, 768, 250112
a = (torch.randn(M, K, device='cuda') * 0.01).to(torch.bfloat16)
w = torch.randn(K, N, device='cuda').to(torch.bfloat16)
out = torch.zeros(M, N, device='cuda')
for name, fn in (('stream-K (atomics)', lambda: ops.gemm_nn_streamk(a, w, out)), ('four-wave', lambda: ops.gemm_nn(a, w, out))):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    m = sorted(ts)[2]
    print('%-22s %8.1f us  %5.0f TF' % (name, m * 1e3, 2.0 * M * N * K / m / 1e9))
