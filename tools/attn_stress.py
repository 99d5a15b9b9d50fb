#!/usr/bin/env python3
"""Attention forward / backward in a loop with freshly allocated operands of changing shapes (the pattern that exposed the
weight-gradient kernel's fragment-read race), every (sequence, head) block against an fp32 torch product on the GPU, dropout
on with the kernel's own keep mask (the RNG twin).  usage: python tools/attn_stress.py [rounds]"""
import math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import ops, rng

SHAPES = [(8, 164, 12, 64), (4, 116, 12, 64), (2, 356, 16, 64), (16, 74, 4, 32), (32, 164, 12, 64), (3, 36, 12, 64), (5, 200, 2, 64)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = n = 0
for rnd in range(rounds):
    for B, S, H, dh in SHAPES:
        d = H * dh
        p = 0.1 if rnd % 2 == 0 else 0.0
        seed = 1000 + rnd
        g = torch.Generator(device='cuda').manual_seed(B * S + rnd)
        qkv = (torch.randn((B * S, 3 * d), device='cuda', generator=g) * 0.7).to(torch.bfloat16)
        dctx = torch.randn((B * S, d), device='cuda', generator=g).to(torch.bfloat16)
        keylen = torch.randint(max(S // 2, 1), S + 1, (B,), device='cuda', dtype=torch.int32, generator=g)
        # (every other dropout round hands the backward the forward's keep-bit words - the training path - instead of the seed)
        km = None
        if p > 0 and rnd % 4 == 0:
            ctx, lse, km = ops.attn_fwd(qkv, keylen, B, S, H, dh, seed=seed, p_drop=p, want_mask=True)
        else:
            ctx, lse = ops.attn_fwd(qkv, keylen, B, S, H, dh, seed=seed, p_drop=p)
        dbias = torch.zeros(3 * d, device='cuda')
        dqkv = ops.attn_bwd(qkv, keylen, ctx, dctx, lse, B, S, H, dh, dbias_qkv=dbias, seed=seed, p_drop=p, keepmask=km)
        x = qkv.float().requires_grad_(True)
        q, k, v = x.view(B, S, 3, H, dh).permute(2, 0, 3, 1, 4)
        sc = q @ k.transpose(2, 3)
        mask = torch.arange(S, device='cuda')[None, :] < keylen[:, None]
        w = torch.softmax(sc.masked_fill(~mask[:, None, None, :], float('-inf')), dim=-1)
        if p > 0:
            keep = torch.from_numpy(rng.keep_mask(B * H * S * S, seed, p, (B, H, S, S))).cuda().float()
            w = w * keep / (1 - p)
        ref = (w @ v).transpose(1, 2).reshape(B * S, d)
        ref.backward(dctx.float())
        gr = x.grad
        gr[:, :d] *= 1.0 / math.sqrt(dh)
        n += 1
        # per (sequence, head) block, rows inside the sequence's own length only (padded query rows carry no gradient)
        rowok = (torch.arange(S, device='cuda')[None, :] >= 0).expand(B, S).reshape(B * S, 1).float()
        e_f = ((ctx.float() - ref.detach()) * rowok).view(B, S, H, dh).pow(2).sum((1, 3)).sqrt() / (ref.detach().view(B, S, H, dh).pow(2).sum((1, 3)).sqrt() + 1e-6)
        e_b = []
        for j in range(3):
            a_, r_ = dqkv[:, j * d:(j + 1) * d].float().view(B, S, H, dh), gr[:, j * d:(j + 1) * d].view(B, S, H, dh)
            e_b.append((a_ - r_).pow(2).sum((1, 3)).sqrt() / (r_.pow(2).sum((1, 3)).sqrt() + 1e-3 * float(r_.norm()) / math.sqrt(B * H)))
        worst = max(float(e_f.max()), *(float(e.max()) for e in e_b))
        if worst > 6e-2:
            bad += 1
            print('BAD round %d B=%d S=%d H=%d dh=%d p=%.1f: worst block fwd %.3e dq %.3e dk %.3e dv %.3e' %
                  (rnd, B, S, H, dh, p, float(e_f.max()), float(e_b[0].max()), float(e_b[1].max()), float(e_b[2].max())), flush=True)
        del qkv, dctx, ctx, dqkv, x, ref
print('%d bad of %d' % (bad, n))
