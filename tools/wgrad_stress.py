#!/usr/bin/env python3
"""Repeats the weight-gradient GEMM on the benchmarked shapes and compares every result with a reference computed once:
hunts intermittent wrong tiles (a race shows up as a few runs with a large error, confined to some tiles).
usage: python tools/wgrad_stress.py [repeats]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
M = 41984
for N, K in ((3072, 768), (2304, 768), (768, 3072), (768, 768)):
    g = torch.Generator(device='cuda').manual_seed(N + K)
    dy = (torch.randn((M, N), device='cuda', generator=g) * 0.1).to(torch.bfloat16)
    x = torch.randn((M, K), device='cuda', generator=g).to(torch.bfloat16)
    ref = dy.float().t() @ x.float() + 1.0
    ref64 = torch.zeros((N, K), dtype=torch.float64, device='cuda')
    for m0 in range(0, M, 8192):
        ref64 += dy[m0:m0 + 8192].double().t() @ x[m0:m0 + 8192].double()
    ref64 += 1.0
    print('  first torch fp32 reference against the fp64 one: rel %.3e' % float((ref.double() - ref64).norm() / ref64.norm()))
    ref = ref64.float()
    bad = 0
    worst = 0.0
    bad_ref = 0
    for r in range(reps):
        dw = torch.ones((N, K), device='cuda')
        ops.gemm_wgrad(dy, x, dw)
        if r % 3 == 0:      # vary what runs beside / before it
            _ = dy.float().sum()
        err = float((dw - ref).norm() / ref.norm())
        if os.environ.get('STRESS_TORCH_REF', '1') != '0':      # is the torch fp32 product (the tests' reference) itself stable?
            r32 = dy.float().t() @ x.float() + 1.0
            e2 = float((r32 - ref).norm() / ref.norm())
            if e2 > 2e-3:
                bad_ref += 1
                print('  N=%d K=%d run %d: the TORCH fp32 reference is off by rel %.3e' % (N, K, r, e2), flush=True)
            del r32
        worst = max(worst, err)
        if err > 2e-3:
            bad += 1
            d = (dw - ref).abs()
            rows = (d.max(dim=1).values > 1e-2).nonzero().view(-1)
            cols = (d.max(dim=0).values > 1e-2).nonzero().view(-1)
            print('  N=%d K=%d run %d: rel %.3e; wrong rows %d..%d (%d), cols %d..%d (%d); max |d| %.3f; mean ratio dw/ref in block %.3f'
                  % (N, K, r, err, int(rows.min()), int(rows.max()), len(rows), int(cols.min()), int(cols.max()), len(cols), float(d.max()),
                     float((dw[rows][:, cols] - 1).sum() / (ref[rows][:, cols] - 1).sum())), flush=True)
    print('N=%d K=%d: ours %d bad of %d (worst rel %.3e); torch fp32 reference %d bad' % (N, K, bad, reps, worst, bad_ref), flush=True)
