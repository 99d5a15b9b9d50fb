#!/usr/bin/env python3
"""Micro-benchmark of the vocabulary cross-entropy kernel at the cfg2 shape (4864 rows x 250 002)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import ops
n, V = 4864, 250002
ld = (V + 63) // 64 * 64
src = (torch.randn(n, ld, device='cuda') * 2).to(torch.bfloat16)
y = torch.randint(0, V, (n,), device='cuda')
logits = src.clone()
for _ in range(2): ops.ce_fwd_bwd(logits.copy_(src), V, y, 1.0 / n, 1.0 / n)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot = 0.0
for _ in range(10):
    logits.copy_(src)
    e0.record(); ops.ce_fwd_bwd(logits, V, y, 1.0 / n, 1.0 / n); e1.record(); torch.cuda.synchronize()
    tot += e0.elapsed_time(e1)
ms = tot / 10
print('ce_fwd_bwd %d x %d: %.3f ms  (%.2f TB/s over 3 passes of %.2f GB)' % (n, V, ms, 3 * n * ld * 2 / ms / 1e9, n * ld * 2 / 1e9))
