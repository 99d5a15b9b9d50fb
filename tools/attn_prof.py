#!/usr/bin/env python3
"""A few attention backward launches at the cfg2 shape for rocprofv3 --pmc runs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import ops
B, S, H, dh = 256, 164, 12, 64
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
torch.manual_seed(0)
qkv = (torch.randn(B * S, 3 * H * dh, device='cuda') * 0.5).to(torch.bfloat16)
keylen = torch.randint(100, S + 1, (B,), device='cuda', dtype=torch.int32)
dctx = (torch.randn(B * S, H * dh, device='cuda') * 0.1).to(torch.bfloat16)
dbias = torch.zeros(3 * H * dh, device='cuda')
ctx, lse, km = ops.attn_fwd(qkv, keylen, B, S, H, dh, seed=5, p_drop=p, want_mask=True)
for _ in range(3):
    ops.attn_bwd(qkv, keylen, ctx, dctx, lse, B, S, H, dh, dbias_qkv=dbias, seed=5, p_drop=p, keepmask=km)
torch.cuda.synchronize()
