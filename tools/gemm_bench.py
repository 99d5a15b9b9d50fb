#!/usr/bin/env python3
"""Micro-benchmark of the GEMM kernels on the cfg2 shapes (used under rocprofv3 for PMC runs).
usage: python tools/gemm_bench.py [nt|wgrad] M N K [iters] [epilogue]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import ops, lib as L   # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else 'nt'
M, N, K = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (41984, 3072, 768)
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
epi = int(sys.argv[6]) if len(sys.argv) > 6 else 0
import ctypes
abl = 0
var = int(os.environ.get('M3P_VARIANT', '1'))
L.load().m3p_debug_set_variant(var)
torch.manual_seed(0)
a = (torch.randn(M, K, device='cuda') * 1.0).to(torch.bfloat16)
w = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
bias = torch.randn(N, device='cuda')
u = torch.empty(M, N, dtype=torch.bfloat16, device='cuda') if epi in (2,) else None
aux = torch.randn(M, N, device='cuda').to(torch.bfloat16) if epi in (3, 4, 5, 6) else None
dw = torch.zeros(N, K, device='cuda')


def run():
    if kind == 'nt':
        ops.gemm_nt(a, w, epi, bias=bias if epi in (1, 2, 3) else None, aux=aux, out=out, out2=u, p_drop=0.1 if epi == 3 else 0.0, seed=3)
    else:
        ops.gemm_wgrad(out, a, dw)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print('variant=%d ablate=%d' % (var, abl), '%s M=%d N=%d K=%d epi=%d: %.4f ms  %.1f TF' % (kind, M, N, K, epi, ms, 2.0 * M * N * K / ms / 1e9))
