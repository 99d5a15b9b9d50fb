#!/usr/bin/env python3
"""The NT GEMM kernels in a loop with freshly allocated operands of changing shapes per call (the pattern that exposed the
weight-gradient kernel's fragment-read race), every element against an fp64 product: a stale or misread fragment shows up as a
block of elements far outside bf16 rounding.  usage: python tools/nt_stress.py [rounds]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import ops, lib as L

SHAPES = [(4096, 768, 768), (4352, 2304, 768), (8192, 3072, 768), (4096, 768, 3072), (41984, 768, 768), (41984, 2304, 768),
          (41984, 768, 3072), (1024, 512, 768), (1280, 2304, 768), (4864, 2048, 768), (96, 768, 768), (32, 3072, 768), (300, 1000, 128),
          (2088, 1000, 128), (22784, 1024, 1024), (22784, 4096, 1024)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = n = 0
for rnd in range(rounds):
    for M, N, K in SHAPES:
        g = torch.Generator(device='cuda').manual_seed(M + N + K + rnd)
        a = torch.randn((M, K), device='cuda', generator=g).to(torch.bfloat16)
        w = (torch.randn((N, K), device='cuda', generator=g) * 0.05).to(torch.bfloat16)
        r = torch.randn((M, N), device='cuda', generator=g).to(torch.bfloat16)
        bias = torch.randn(N, device='cuda', generator=g)
        ref = a.double() @ w.double().t()
        for epi, kw, refe in ((L.EPI_NONE, {}, ref), (L.EPI_BIAS, dict(bias=bias), ref + bias.double()), (L.EPI_RES, dict(aux=r), ref + r.double())):
            if epi != L.EPI_NONE and rnd % 3 != 0:
                continue
            c = ops.gemm_nt(a, w, epi, **kw)
            err = (c.double() - refe).abs()
            tol = 0.012 * refe.abs() + 0.02 * float(refe.abs().mean())      # ~3 bf16 ulps + an absolute floor
            n += 1
            nb = int((err > tol).sum())
            if nb:
                bad += 1
                idx = (err > tol).nonzero()
                print('BAD round %d M=%d N=%d K=%d epi %d: %d elements; rows %d..%d cols %d..%d; worst %.3f (ref %.3f)'
                      % (rnd, M, N, K, epi, nb, int(idx[:, 0].min()), int(idx[:, 0].max()), int(idx[:, 1].min()), int(idx[:, 1].max()),
                         float(err.max()), float(refe.abs().mean())), flush=True)
            del c, err, tol
        del a, w, r, ref
print('%d bad of %d' % (bad, n))
