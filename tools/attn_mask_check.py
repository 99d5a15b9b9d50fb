#!/usr/bin/env python3
"""The keep words the attention forward records (word [head][qb][t][r], bit l = keep(query 16 qb + (l & 15), key 16 t + 4 (l >> 4) + r))
against the NumPy twin of the dropout stream, and the two backward forms (re-hashed / fed the words) against each other."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import ops, rng
B, S, H, dh = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (3, 164, 12, 64)
p, seed = 0.1, 4242
d = H * dh
torch.manual_seed(1)
qkv = (torch.randn(B * S, 3 * d, device='cuda') * 0.7).to(torch.bfloat16)
rs = np.random.RandomState(3); kl = rs.randint(max(S // 2, 1), S + 1, size=B).astype(np.int32); kl[0] = S; keylen = torch.from_numpy(kl).cuda()
ctx, lse, km = ops.attn_fwd(qkv, keylen, B, S, H, dh, seed=seed, p_drop=p, want_mask=True)
nt = (S + 15) // 16
w = km.cpu().numpy().view(np.uint64).reshape(B * H, nt, nt, 4)
twin = rng.keep_mask(B * H * S * S, seed, p, (B * H, S, S))
bad = 0
for qb in range(nt):
    for t in range(nt):
        for r in range(4):
            word = w[:, qb, t, r]
            for l in range(64):
                q, k = 16 * qb + (l & 15), 16 * t + 4 * (l >> 4) + r
                if q < S and k < S:
                    bit = (word >> np.uint64(l)) & np.uint64(1)
                    bad += int((bit.astype(bool) != twin[:, q, k]).sum())
print('keep words vs twin: %d mismatching bits of %d' % (bad, B * H * S * S))
dctx = torch.randn(B * S, d, device='cuda').to(torch.bfloat16)
a = ops.attn_bwd(qkv, keylen, ctx, dctx, lse, B, S, H, dh, dbias_qkv=torch.zeros(3 * d, device='cuda'), seed=seed, p_drop=p)
b = ops.attn_bwd(qkv, keylen, ctx, dctx, lse, B, S, H, dh, dbias_qkv=torch.zeros(3 * d, device='cuda'), seed=seed, p_drop=p, keepmask=km)
diff = (a.float() - b.float()).abs()
print('re-hashed vs fed: %d differing elements, max %.4g; by column block (dq, dk, dv): %s' %
      (int((diff > 0).sum()), float(diff.max()), [int((diff[:, i * d:(i + 1) * d] > 0).sum()) for i in range(3)]))
rows = (diff > 0).any(1).nonzero().flatten()
print('keylen', kl.tolist())
print('rows with differences (batch, position):', [(int(r) // S, int(r) % S) for r in rows[:40]], '... of', rows.numel())
cols = (diff > 0).any(0).nonzero().flatten()
print('columns:', cols[:40].tolist(), '... of', cols.numel())
