#!/usr/bin/env python3
"""Which kernels the vendor library picks for the long-K layer products (run under rocprofv3 --kernel-trace --stats)."""
import torch
M = 41984
for N, K in ((768, 3072), (768, 2304), (3072, 768), (768, 768), (2304, 768)):
    a = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    w = torch.randn(N, K, device='cuda').to(torch.bfloat16)
    for _ in range(5):
        c = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        c = a @ w.t()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print('M=%d N=%d K=%d: %.1f us %.0f TF' % (M, N, K, t * 1e3, 2 * M * N * K / t / 1e9))
