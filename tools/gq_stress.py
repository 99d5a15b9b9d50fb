#!/usr/bin/env python3
"""The byte epilogues of the eight-wave kernel (lin1 + GELU + byte from the LDS table, byte-decode dU; both with and without the
8-bit copy of their output) in a loop with freshly allocated operands of changing shapes per call, every element against an
fp64 product: h against erf-GELU of the fp64 pre-activation, the decoded codes against the exact derivative, dU against
product x decoded code, the column sums, the 8-bit copies against the OCP casts.   usage: python tools/gq_stress.py [rounds]"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import ops, lib as L

SHAPES = [(1024, 512, 64), (2048, 768, 256), (4096, 3072, 768), (1280, 2304, 128), (41984, 3072, 768), (22784, 4096, 1024), (8192, 1024, 1024)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
bad = n = 0


def report(what, err, tol, M, N, K, rnd):
    global bad
    nb = int((err > tol).sum())
    if nb:
        bad += 1
        idx = (err > tol).nonzero()
        print('BAD %s round %d M=%d N=%d K=%d: %d elements; rows %d..%d cols %d..%d; worst %.4g' %
              (what, rnd, M, N, K, nb, int(idx[:, 0].min()), int(idx[:, 0].max()), int(idx[:, -1].min()), int(idx[:, -1].max()), float(err.max())), flush=True)


for rnd in range(rounds):
    for M, N, K in SHAPES:
        if M * N > 60e6 and rnd % 3:
            continue
        g = torch.Generator(device='cuda').manual_seed(M + N + K + rnd)
        a = torch.randn((M, K), device='cuda', generator=g).to(torch.bfloat16)
        w = (torch.randn((N, K), device='cuda', generator=g) * (1.4 / math.sqrt(K))).to(torch.bfloat16)
        bias = torch.randn(N, device='cuda', generator=g) * 0.5
        with8 = (rnd + M) % 2 == 0
        q = torch.empty(M * N, dtype=torch.uint8, device='cuda')
        kw = {}
        if with8:
            kw = dict(out8=torch.empty((M, N), dtype=torch.uint8, device='cuda'), scale8=torch.tensor([20.0], device='cuda'),
                      amax8=torch.zeros(1, device='cuda'))
        h = ops.gemm_nt(a, w, L.EPI_BIAS_GELUQ, bias=bias, out2=q, **kw)
        n += 1
        for r0 in range(0, M, 8192):          # fp64 reference in row slices
            sl = slice(r0, min(r0 + 8192, M))
            u = a[sl].double() @ w.double().t() + bias.double()
            Phi = 0.5 * (1 + torch.erf(u / math.sqrt(2)))
            href = u * Phi
            report('gelu', (h[sl].double() - href).abs(), 0.012 * href.abs() + 3e-3, M, N, K, rnd)
            dref = Phi + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi)
            dec = ops.gq_unpack(q, M, N)[sl].double()
            # (the kernel's u is its own fp32 accumulation, rounded to bf16 precision for the table index: 1.5 code steps + the index rounding)
            report('code', (dec - dref).abs(), torch.full_like(dref, 1.5 * ops.GQ_STEP) + 0.8 * u.abs() * 2.0 ** -8, M, N, K, rnd)
            if with8:
                got = kw['out8'][sl].view(torch.float8_e4m3fn).double()
                ref8 = (href * 20.0).clamp(-448, 448)
                report('h8', (got - ref8).abs(), 0.07 * ref8.abs() + 0.05, M, N, K, rnd)
            del u, Phi, href, dref, dec
        # backward through the codes
        dy = (torch.randn((M, K), device='cuda', generator=g) * 0.05).to(torch.bfloat16)
        cs = torch.zeros(N, device='cuda')
        kw2 = {}
        if with8:
            kw2 = dict(out8=torch.empty((M, N), dtype=torch.uint8, device='cuda'), scale8=torch.tensor([64.0], device='cuda'),
                       amax8=torch.zeros(1, device='cuda'), out8_bf8=True)
        du = ops.gemm_nt(dy, w, L.EPI_MULQ, aux=q, colsum=cs, **kw2)
        n += 1
        csr = torch.zeros(N, dtype=torch.float64, device='cuda')
        for r0 in range(0, M, 8192):
            sl = slice(r0, min(r0 + 8192, M))
            ref = (dy[sl].double() @ w.double().t()) * ops.gq_unpack(q, M, N)[sl].double()
            report('mulq', (du[sl].double() - ref).abs(), 0.012 * ref.abs() + 0.02 * float(ref.abs().mean()), M, N, K, rnd)
            csr += ref.sum(0)
            if with8:
                got = kw2['out8'][sl].view(torch.float8_e5m2).double()
                report('du8', (got - ref * 64.0).abs(), 0.14 * (ref * 64.0).abs() + 64.0 * 0.02 * float(ref.abs().mean()), M, N, K, rnd)
            del ref
        if float((cs.double() - csr).norm() / csr.norm()) > 2e-2:
            bad += 1
            print('BAD colsum round %d M=%d N=%d K=%d' % (rnd, M, N, K), flush=True)
        del a, w, h, q, du, dy
print('%d bad of %d' % (bad, n))
