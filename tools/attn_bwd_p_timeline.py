#!/usr/bin/env python3
"""Per-segment s_memtime sums of the PERSISTENT one-pass attention backward (needs a -DM3P_ATTN_TL build:
tools/build_alt.sh -DM3P_ATTN_TL, M3P_HIP_LIB=m3p_amd/libm3p_hip_alt.so): ticks per head and segment, waves 0-3 of every workgroup."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import lib as L, ops
B, S, H, dh = 256, 164, 12, 64
lib = L.load()
f = lib.m3p_debug_attn_timeline
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_size_t]
torch.manual_seed(0)
qkv = (torch.randn(B * S, 3 * H * dh, device='cuda') * 0.5).to(torch.bfloat16)
keylen = torch.randint(100, S + 1, (B,), device='cuda', dtype=torch.int32)
ctx, lse, mask = ops.attn_fwd(qkv, keylen, B, S, H, dh, seed=5, p_drop=0.1, want_mask=True)
dctx = torch.randn_like(ctx)
dbias = torch.zeros(3 * H * dh, device='cuda')
for _ in range(3):
    ops.attn_bwd(qkv, keylen, ctx, dctx, lse, B, S, H, dh, dbias_qkv=dbias, seed=5, p_drop=0.1, keepmask=mask)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.attn_bwd(qkv, keylen, ctx, dctx, lse, B, S, H, dh, dbias_qkv=dbias, seed=5, p_drop=0.1, keepmask=mask)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
buf = np.zeros((4096, 4, 16), dtype=np.uint64)
assert f(buf.ctypes.data, buf.nbytes) == 0
ncu = torch.cuda.get_device_properties(0).multi_processor_count
t = buf[:ncu, :, :8].astype(np.float64)
heads = t[..., 7]
names = ['wait for the head\'s requests', 'D / lse + barrier 1', 'previous head: bias sums, stores; K request', 'phase A', 'bias flush + barrier 2',
         'next head\'s requests', 'phase B']
print('persistent one-pass attention backward B=%d S=%d H=%d p=0.1: %.1f us (instrumented build), %.1f heads per workgroup' % (B, S, H, us, heads.mean()))
tot = t[..., :7].sum(-1) / heads
for k, nm in enumerate(names):
    x = (t[..., k] / heads).ravel()
    print('  %-46s %8.0f ticks per head  (p10 %6.0f  p90 %6.0f)  %5.1f %%' % (nm, x.mean(), np.percentile(x, 10), np.percentile(x, 90), 100 * x.mean() / tot.mean()))
print('  %-46s %8.0f' % ('per head', tot.mean()))
