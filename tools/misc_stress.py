#!/usr/bin/env python3
"""Fresh-operand stress of the remaining LDS-DMA GEMM kernels: the stream-K vocabulary data gradient (fp32 atomics) against an
fp64 product, and the fp8 eight-wave kernel against the exact product of its own 8-bit operands (dequantised on the GPU).
usage: python tools/misc_stress.py [rounds]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import ops, lib as L

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = n = 0
for rnd in range(rounds):
    # ---- stream-K: out[M, N] += a[M, K] @ w[K, N]
    for M, N, K in ((4864, 768, 25024), (1216, 768, 100096), (19456, 768, 12544), (600, 128, 25024)):
        g = torch.Generator(device='cuda').manual_seed(M + K + rnd)
        a = (torch.randn((M, K), device='cuda', generator=g) * 0.05).to(torch.bfloat16)
        w = (torch.randn((K, N), device='cuda', generator=g) * 0.05).to(torch.bfloat16)
        out = torch.ones((M, N), device='cuda')
        ops.gemm_nn_streamk(a, w, out, alpha=0.5)
        ref = torch.ones((M, N), dtype=torch.float64, device='cuda')
        for k0 in range(0, K, 16384):
            ref += 0.5 * (a[:, k0:k0 + 16384].double() @ w[k0:k0 + 16384].double())
        err = float((out.double() - ref).norm() / ref.norm())
        n += 1
        if err > 1e-5:
            bad += 1
            d = (out.double() - ref).abs()
            print('BAD stream-K round %d M=%d N=%d K=%d rel %.3e; rows %d..%d' % (rnd, M, N, K, err, int((d.max(1).values > 1e-3).nonzero().min()),
                                                                                     int((d.max(1).values > 1e-3).nonzero().max())), flush=True)
        del a, w, out, ref
    # ---- fp8: C = (a8 / sa) (w8 / sw)^T
    for M, N, K in ((4096, 1024, 1024), (22784, 4096, 1024), (22784, 1024, 4096), (8192, 3072, 1024)):
        g = torch.Generator(device='cuda').manual_seed(M + N + rnd)
        a = torch.randn((M, K), device='cuda', generator=g).to(torch.bfloat16)
        w = (torch.randn((N, K), device='cuda', generator=g) * 0.05).to(torch.bfloat16)
        sa, sw = torch.tensor([32.0], device='cuda'), torch.tensor([512.0], device='cuda')
        a8, w8 = ops.quant_fp8(a, scale=sa), ops.quant_fp8(w, scale=sw)
        c = ops.gemm_nt_fp8(a8, w8, L.EPI_NONE, descale_a=1.0 / sa, descale_b=1.0 / sw)
        ref = (a8.view(torch.float8_e4m3fn).double() / 32.0) @ (w8.view(torch.float8_e4m3fn).double() / 512.0).t()
        e = (c.double() - ref).abs()
        tol = 0.012 * ref.abs() + 0.02 * float(ref.abs().mean())
        n += 1
        if int((e > tol).sum()):
            bad += 1
            idx = (e > tol).nonzero()
            print('BAD fp8 round %d M=%d N=%d K=%d: %d elements; rows %d..%d cols %d..%d' % (rnd, M, N, K, len(idx), int(idx[:, 0].min()),
                                                                                             int(idx[:, 0].max()), int(idx[:, 1].min()), int(idx[:, 1].max())), flush=True)
        del a, w, a8, w8, c, ref
print('%d bad of %d' % (bad, n))
