#!/usr/bin/env python3
"""fp8 NT GEMM against the bf16 kernel on the M3P-large layer shapes (configs[3]: d = 1024, M = 64 x 356) and the cfg2 ones."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import ops, lib as L


def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (M, N, K) in [(22784 // 256 * 256, 3072, 1024), (22784 // 256 * 256, 1024, 1024), (22784 // 256 * 256, 4096, 1024), (22784 // 256 * 256, 1024, 4096),
                  (41984, 2304, 768), (41984, 3072, 768), (41984, 768, 3072), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    w = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
    one = torch.ones(1, device='cuda')
    a8, w8 = ops.quant_fp8(a), ops.quant_fp8(w, scale=one * 16)
    ms16 = t(lambda: ops.gemm_nt(a, w, 0, out=out))
    ms8 = t(lambda: ops.gemm_nt_fp8(a8, w8, 0, descale_a=one, descale_b=one, out=out))
    msq = t(lambda: ops.quant_fp8(a, scale=one))
    f = 2.0 * M * N * K / 1e9
    print('M=%d N=%d K=%d  bf16 %.1f us %.0f TF | fp8 %.1f us %.0f TF (x%.2f) | quant A %.1f us' % (M, N, K, ms16 * 1e3, f / ms16, ms8 * 1e3, f / ms8, ms16 / ms8, msq * 1e3))
