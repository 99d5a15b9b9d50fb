#!/usr/bin/env python3
"""Micro-benchmark of the attention kernels: attn_bench.py [B S H dh] (default: the cfg2 shape 256 164 12 64)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import ops
B, S, H, dh = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (256, 164, 12, 64)
torch.manual_seed(0)
qkv = (torch.randn(B * S, 3 * H * dh, device='cuda') * 0.5).to(torch.bfloat16)
keylen = torch.randint(S // 2, S + 1, (B,), device='cuda', dtype=torch.int32)
dctx = (torch.randn(B * S, H * dh, device='cuda') * 0.1).to(torch.bfloat16)
dbias = torch.zeros(3 * H * dh, device='cuda')

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000

for p in (0.1, 0.0):
    ctx, lse, km = ops.attn_fwd(qkv, keylen, B, S, H, dh, seed=5, p_drop=p, want_mask=True)
    us_f = t(lambda: ops.attn_fwd(qkv, keylen, B, S, H, dh, seed=5, p_drop=p))
    us_fm = t(lambda: ops.attn_fwd(qkv, keylen, B, S, H, dh, seed=5, p_drop=p, want_mask=True))
    us_bm = t(lambda: ops.attn_bwd(qkv, keylen, ctx, dctx, lse, B, S, H, dh, dbias_qkv=dbias, seed=5, p_drop=p, keepmask=km))
    us_b0 = t(lambda: ops.attn_bwd(qkv, keylen, ctx, dctx, lse, B, S, H, dh, dbias_qkv=None, seed=5, p_drop=p, keepmask=km))
    pairs = B * H * S * S
    print('B=%d S=%d H=%d dh=%d  fwd %.2f ps/pair  bwd %.2f ps/pair' % (B, S, H, dh, us_fm * 1e6 / pairs, us_bm * 1e6 / pairs))
    print('p_drop=%.1f  fwd %.1f us (with mask out %.1f)   bwd %.1f us   bwd(no dbias) %.1f us' % (p, us_f, us_fm, us_bm, us_b0), flush=True)
