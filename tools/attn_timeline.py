#!/usr/bin/env python3
"""Phase timeline of the attention forward kernel (needs a -DM3P_ATTN_TL build: tools/build_alt.sh -DM3P_ATTN_TL,
M3P_HIP_LIB=m3p_amd/libm3p_hip_alt.so).  Prints s_memtime deltas of the first query block of every wave and the
workgroup residency derived from s_memrealtime (100 MHz)."""
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
for _ in range(3):
    ops.attn_fwd(qkv, keylen, B, S, H, dh, seed=5, p_drop=p, want_mask=True)
torch.cuda.synchronize()
n = min(B * H, 4096)
buf = np.zeros((4096, 4, 16), dtype=np.uint64)
rc = f(buf.ctypes.data, buf.nbytes); assert rc == 0, rc
t = buf[:n].astype(np.int64)
names = ['stage issue', 'stage wait+barrier', 'Q load + QK^T', 'softmax', 'dropout+pack', 'P V', 'store', 'other q-blocks']
d = np.diff(t[:, :, 0:9], axis=-1)
print('cycles (s_memtime) per phase, mean / p10 / p90 over %d workgroups x 4 waves' % n)
for k, nm in enumerate(names):
    x = d[..., k].ravel()
    print('  %-22s %8.0f %8.0f %8.0f' % (nm, x.mean(), np.percentile(x, 10), np.percentile(x, 90)))
tot = (t[:, :, 8] - t[:, :, 0])
print('  wave total             %8.0f   per wave id: %s' % (tot.mean(), np.round(tot.mean(0)).tolist()))
rt0, rt1 = t[:, :, 14].min(1), t[:, :, 15].max(1)
life = (rt1 - rt0) / 100.0
span = (rt1.max() - rt0.min()) / 100.0
print('workgroup residency us: mean %.1f p10 %.1f p90 %.1f; kernel span %.1f us; mean concurrency %.0f WGs (%.2f per CU)'
      % (life.mean(), np.percentile(life, 10), np.percentile(life, 90), span, life.sum() / span, life.sum() / span / 256))
print('shader clock estimate: %.2f GHz' % (tot.mean() / ((t[:, :, 15] - t[:, :, 14]).mean() * 10.0)))
