#!/usr/bin/env python3
"""Phase timeline of the attention BACKWARD kernel (needs a -DM3P_ATTN_TL build: tools/build_alt.sh -DM3P_ATTN_TL,
M3P_HIP_LIB=m3p_amd/libm3p_hip_alt.so): s_memtime stamps per wave, see the BTL comment in csrc/attention.hip."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import lib as L, ops
B, S, H, dh = 256, 164, 12, 64
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
lib = L.load()
f = lib.m3p_debug_attn_timeline
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_size_t]
torch.manual_seed(0)
qkv = (torch.randn(B * S, 3 * H * dh, device='cuda') * 0.5).to(torch.bfloat16)
keylen = torch.randint(100, S + 1, (B,), device='cuda', dtype=torch.int32)
ctx, lse, mask = ops.attn_fwd(qkv, keylen, B, S, H, dh, seed=5, p_drop=p, want_mask=True)
dctx = torch.randn_like(ctx)
dbias = torch.zeros(3 * H * dh, device='cuda')
for _ in range(3):
    ops.attn_bwd(qkv, keylen, ctx, dctx, lse, B, S, H, dh, dbias_qkv=dbias, seed=5, p_drop=p, keepmask=mask)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.attn_bwd(qkv, keylen, ctx, dctx, lse, B, S, H, dh, dbias_qkv=dbias, seed=5, p_drop=p, keepmask=mask)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
n = min(B * H, 4096)
buf = np.zeros((4096, 4, 16), dtype=np.uint64)
rc = f(buf.ctypes.data, buf.nbytes); assert rc == 0, rc
t = buf[:n].astype(np.int64)
names = ['D = rowsum(dO O), lse', 'stage Q, dO + barrier', 'phase A (dV, dK)', 'bias flush + barrier', 'stage K, V + barrier',
         'phase B (dQ)', 'bias reduce + end']
d = np.diff(t[:, :, 0:8], axis=-1)
tot = t[:, :, 7] - t[:, :, 0]
print('attention backward B=%d S=%d H=%d p=%.2f: %.1f us (instrumented build)' % (B, S, H, p, us))
print('cycles (s_memtime) per segment, mean / p10 / p90 over %d workgroups x 4 waves; share of the wave total' % n)
for k, nm in enumerate(names):
    x = d[..., k].ravel()
    print('  %-26s %8.0f %8.0f %8.0f   %5.1f %%   per wave id %s' % (nm, x.mean(), np.percentile(x, 10), np.percentile(x, 90), 100 * x.mean() / tot.mean(),
                                                                  np.round(d[..., k].mean(0)).astype(int).tolist()))
print('  wave total                 %8.0f' % tot.mean())
for k, nm in ((8, 'A: wait for owned K / V rows'), (9, 'B: wait for owned Q / dO rows'), (10, 'A: output stores'), (11, 'B: output stores')):
    x = t[:, :, k].ravel()
    print('  of which %-28s %8.0f  (%4.1f %%)' % (nm, x.mean(), 100 * x.mean() / tot.mean()))
for k, nm in ((12, 'A step: row fragments + score / dPd MFMAs issued'), (13, 'A step: P / dS arithmetic (incl. waiting for the MFMAs, keep words)'), (14, 'A step: transposed fragments + dV / dK MFMAs issued')):
    x = t[:, :, k].ravel()
    print('  phase A steps, %-70s %8.0f  (%4.1f %% of phase A)' % (nm, x.mean(), 100 * x.mean() / d[..., 2].mean()))
# workgroup residency against the launch: 3072 workgroups, 768 resident at a time if three fit a CU
span = t[:, :, 7].max() - t[:, :, 0].min()
life = t[:, :, 7].max(1) - t[:, :, 0].min(1)
print('workgroup residency: mean %.0f ticks; launch span %.0f ticks (%.2f GHz if the span is the %.1f us); mean concurrency %.0f workgroups (%.2f per CU)'
      % (life.mean(), span, span / us / 1e3, us, life.sum() / span, life.sum() / span / 256))
