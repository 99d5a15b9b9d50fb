#!/usr/bin/env python3
"""Per-wave s_memtime breakdown of the persistent stream-K weight-gradient kernel (debug hook)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import lib as L
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (41984, 3072, 768)
lib = L.load()
f = lib.m3p_debug_wgrad_timeline
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
dy = torch.randn(M, N, device='cuda').to(torch.bfloat16); x = torch.randn(M, K, device='cuda').to(torch.bfloat16)
dw = torch.zeros(N, K, device='cuda')
dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device='cuda')
for _ in range(3):
    rc = f(dy.data_ptr(), N, x.data_ptr(), K, dw.data_ptr(), K, M, N, K, dbg.data_ptr(), L.stream())
torch.cuda.synchronize()
d = dbg.view(256, 8, 8).double()
names = ['issue loads', 'tr reads + mfma batch1', 'lgkm wait', 'vmcnt wait', 'barrier', 'tr reads+mfma batch2+lgkm', 'flush/loop tail', 'loop head']
tot = d.sum(-1).mean()
print('shape', M, N, K, ' mean cycles per wave: %.0f' % tot)
for k, n in enumerate(names):
    print('%-28s mean %9.0f (%.1f%%)  min %9.0f max %9.0f' % (n, d[..., k].mean(), 100 * d[..., k].mean() / tot, d[..., k].min(), d[..., k].max()))
