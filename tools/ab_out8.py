#!/usr/bin/env python3
"""What the 8-bit copy costs the two byte epilogues (M3PEpilogue::out8) at a given shape: each with and without it, alternating."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import ops, lib as L
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (22784, 4096, 1024)
a = torch.randn(M, K, device='cuda').to(torch.bfloat16); w = (torch.randn(N, K, device='cuda') * (1.4 / math.sqrt(K))).to(torch.bfloat16)
bias = torch.randn(N, device='cuda'); q = torch.empty(M * N, dtype=torch.uint8, device='cuda'); o8 = torch.empty(M, N, dtype=torch.uint8, device='cuda')
sc = torch.tensor([16.0], device='cuda'); am = torch.zeros(1, device='cuda'); cs = torch.zeros(N, device='cuda'); out = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1000
arms = {'geluq': lambda: ops.gemm_nt(a, w, L.EPI_BIAS_GELUQ, bias=bias, out2=q, out=out),
        'geluq + e4m3 copy': lambda: ops.gemm_nt(a, w, L.EPI_BIAS_GELUQ, bias=bias, out2=q, out=out, out8=o8, scale8=sc, amax8=am),
        'mulq': lambda: ops.gemm_nt(a, w, L.EPI_MULQ, aux=q, colsum=cs, out=out),
        'mulq + e5m2 copy': lambda: ops.gemm_nt(a, w, L.EPI_MULQ, aux=q, colsum=cs, out=out, out8=o8, scale8=sc, amax8=am, out8_bf8=True)}
res = {k: [] for k in arms}
for r in range(5):
    for k, fn in arms.items(): res[k].append(t(fn))
for k, v in res.items(): print('%-20s %7.1f us  (M=%d N=%d K=%d)' % (k, sorted(v)[2], M, N, K))
