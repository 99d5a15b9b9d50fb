#!/usr/bin/env python3
"""Per-wave s_memtime breakdown of the four-wave NT GEMM (debug instantiation; M3P_VARIANT high byte = ablation:
256 no fragment reads, 512 no LDS-DMA, 1024 every K-tile re-reads the first)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import lib as L
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (41984, 768, 3072)
lib = L.load()
f = lib.m3p_debug_gemm_timeline
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
a = torch.randn(M, K, device='cuda').to(torch.bfloat16); w = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device='cuda')
for _ in range(3):
    rc = f(a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, dbg.data_ptr(), L.stream())
torch.cuda.synchronize()
d = dbg.view(256, 8, 8).double()
d = d[:, :4]       # four waves per workgroup
names = ['groups (reads+dma+mfma issue)', 'lgkm wait', 'vmcnt wait', 'barrier', 'epilogue', 'loop glue', 'drain', '-']
if os.environ.get('W4_SCHED2', '1') == '1':      # the one-piece K-tile: MFMA 0..20 + barrier 1 | ..50 + barrier 2 | ..107 | vmcnt wait | barrier 3 | ..127 + lgkmcnt(0)
    names = ['MFMA 0-20, W reads, barrier 1', 'MFMA 21-50, A reads, W DMAs, barrier 2', 'MFMA 51-107, A DMAs', 'vmcnt(16) wait', 'barrier 3', 'MFMA 108-127, 16 reads, lgkmcnt(0)', 'epilogue', 'drain']
tot = d.sum(-1).mean()
print('shape', M, N, K, ' mean cycles per wave: %.0f' % tot)
for k, n in enumerate(names):
    print('%-28s mean %9.0f (%.1f%%)  min %9.0f max %9.0f' % (n, d[..., k].mean(), 100 * d[..., k].mean() / tot, d[..., k].min(), d[..., k].max()))
