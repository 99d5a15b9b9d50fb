#!/usr/bin/env python3
"""Where the eight-wave NT kernel's dGELU launch spends its cycles (needs tools/build_alt.sh -DM3P_RING_TL and
M3P_HIP_LIB=m3p_amd/libm3p_hip_alt.so): per-wave s_memtime sums per segment, with / without the column sums."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import lib as L, ops
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (41984, 3072, 768)
lib = L.load()
f = lib.m3p_debug_ring_timeline
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_size_t]
a = (torch.randn(M, K, device='cuda')).to(torch.bfloat16); w = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
aux = torch.randn(M, N, device='cuda').to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
cs = torch.zeros(N, device='cuda')
names = ['K loop', 'bias + row copies', 'aux fetch + wait', 'epilogue half (compute, staging, stores)', 'column sums', 'zero + end barrier']
for colsum in (None, cs):
    for _ in range(3):
        ops.gemm_nt(a, w, L.EPI_DGELU, aux=aux, out=out, colsum=colsum)
    torch.cuda.synchronize()
    buf = np.zeros((256, 8, 8), dtype=np.uint64)
    rc = f(buf.ctypes.data, buf.nbytes); assert rc == 0, rc
    d = buf.astype(np.float64)
    tot = d.sum(-1).mean()
    print('dGELU M=%d N=%d K=%d colsum=%s: mean cycles per wave %.0f' % (M, N, K, colsum is not None, tot))
    for k, nm in enumerate(names):
        print('  %-42s %9.0f (%.1f%%)' % (nm, d[..., k].mean(), 100 * d[..., k].mean() / tot))
