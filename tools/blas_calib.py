#!/usr/bin/env python3
"""Calibration only (not a product path): what does the vendor GEMM reach on the cfg2 shapes on
this box?  Prints TF/s for torch.matmul (hipBLASLt) next to our ring kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import ops

def t(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for (M, N, K) in [(41984, 3072, 768), (41984, 768, 3072), (41984, 2304, 768), (41984, 768, 768), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    w = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
    ms_b = t(lambda: torch.matmul(a, w.t(), out=out))
    ms_o = t(lambda: ops.gemm_nt(a, w, 0, out=out))
    # wgrad shape: dW[N,K] = dY[M,N]^T X[M,K]
    dy = torch.randn(M, N, device='cuda').to(torch.bfloat16)
    dw = torch.zeros(N, K, device='cuda')
    dwb = torch.empty(N, K, dtype=torch.bfloat16, device='cuda')
    ms_wb = t(lambda: torch.matmul(dy.t(), a, out=dwb))
    ms_wo = t(lambda: ops.gemm_wgrad(dy, a, dw))
    f = 2.0 * M * N * K / 1e9
    print('M=%d N=%d K=%d  NT: blas %.1f TF  ours %.1f TF | wgrad: blas %.1f TF  ours %.1f TF' % (M, N, K, f / ms_b, f / ms_o, f / ms_wb, f / ms_wo))
