#!/usr/bin/env python3
"""Time of the cross-entropy gradient pass at the benchmarked size (4864 x 250 002 logits): M3P_HIP_LIB selects the build."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import ops
M, V, ld = 4864, 250002, 250112
base = (torch.randn(M, ld, device='cuda') * 2).to(torch.bfloat16)
tgt = torch.randint(4, V - 2, (M,), device='cuda')
ts = []
for it in range(6):
    logits = base.clone()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = ops.ce_fwd_bwd_colsum(logits[:, :V] if False else logits, V, tgt, 1.0, 1.0 / M)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print(os.environ.get('M3P_HIP_LIB', 'default'), 'ce_fwd_bwd_colsum: %.1f us (min of %d)' % (min(ts[1:]) * 1e3, len(ts) - 1), 'checksum %.6f' % float(logits.float().abs().sum()))
