#!/usr/bin/env python3
"""A/B of NT GEMM kernels inside ONE process: every (library build, M3P_VARIANT) arm is a separate CDLL handle, the arms
alternate round-robin on the same operands and the median of several rounds is reported - box-to-box (and
minute-to-minute clock) spread is +-6 %, far more than the differences being measured.

    python tools/ab_gemm.py libm3p_hip.so:1 libm3p_hip.so:6 libm3p_hip_alt.so:6      # <file under m3p_amd/>:<variant>
"""
import ctypes as C
import os
import shutil
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m3p_amd import lib as L   # noqa: E402

arms = []
tmp = tempfile.mkdtemp()
for k, spec in enumerate(sys.argv[1:]):
    name, var = spec.split(':')
    # a private copy per arm: dlopen of the same path would return the same handle (and the same variant switch)
    path = os.path.join(tmp, 'arm%d.so' % k)
    shutil.copy(os.path.join(ROOT, 'm3p_amd', name), path)
    h = C.CDLL(path)
    fn = h.m3p_gemm_nt_bf16
    fn.restype, fn.argtypes = L.SIGNATURES['m3p_gemm_nt_bf16']
    h.m3p_debug_set_variant(int(var))
    if os.environ.get('AB_GRID'):
        assert h.m3p_set_persistent_grid(int(os.environ['AB_GRID'])) == 0
    arms.append((spec, fn))

SHAPES = [('QKV fwd', 2304, 768, 1), ('FFN1 fwd', 3072, 768, 1), ('out_lin fwd', 768, 768, 3), ('FFN2 fwd', 768, 3072, 3),
          ('vocab', 250112, 768, 1), ('vocab lse', 250112, 768, 9), ('dx1', 768, 3072, 4), ('dh', 768, 2304, 4), ('dctx', 768, 768, 0), ('FFN1 gelu', 3072, 768, 2), ('FFN1 geluq', 3072, 768, 8), ('dU dgelu', 3072, 768, 5), ('dU mul', 3072, 768, 6), ('dU mulq', 3072, 768, 7)]
M = int(os.environ.get('AB_M', '41984'))
if os.environ.get('AB_ONLY'):      # comma-separated substrings of the shape names to keep
    SHAPES = [sh for sh in SHAPES if any(k in sh[0] for k in os.environ['AB_ONLY'].split(','))]
st = torch.cuda.current_stream().cuda_stream
print('%-14s' % 'shape' + ''.join('%22s' % a[0] for a in arms))
tot = [0.0] * len(arms)
M0 = M
for name, N, K, epi in SHAPES:
    M = 4864 if name.startswith('vocab') else M0
    a = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    w = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
    bias = torch.randn(N, device='cuda')
    aux = torch.randn(M, N, device='cuda').to(torch.bfloat16)
    cs = torch.zeros(N, device='cuda')
    ep = L.Epilogue()
    ep.bias = bias.data_ptr() if epi in (1, 2, 3, 8, 9) else None
    stats = torch.empty((N // 64, M, 2), device='cuda') if epi == 9 else None
    auxq = torch.randint(0, 256, (M * N,), dtype=torch.uint8, device='cuda') if epi in (7, 8) else None
    ep.aux = aux.data_ptr() if epi in (3, 4, 5, 6) else (auxq.data_ptr() if epi == 7 else None)
    ep.colsum = cs.data_ptr() if epi in (5, 6, 7) else None
    ep.ld_aux = N
    ep.out2 = aux.data_ptr() if epi == 2 else (auxq.data_ptr() if epi == 8 else (stats.data_ptr() if epi == 9 else None))
    ep.ld_out2 = N - 110 if epi == 9 else N
    ep.alpha = 1.0
    ep.seed = 3
    ep.thresh24 = L.thresh24(0.1) if epi == 3 else 0
    ep.inv_keep = 1.0 / 0.9 if epi == 3 else 1.0

    def run(fn, n):
        for _ in range(n):
            rc = fn(a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, epi, C.byref(ep), st)
            if rc == -1 and epi in (7, 8, 9):
                return False          # (an arm built before round 4: no byte-derivative epilogue)
            assert rc == 0, rc
        return True
    times = [[] for _ in arms]
    ok = [run(fn[1], 5) for fn in arms]
    for rnd in range(7):
        for i, (_, fn) in enumerate(arms):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(fn, 10)
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / 10 if ok[i] else float('nan'))
    med = [sorted(t)[len(t) // 2] for t in times]
    for i, m in enumerate(med):
        if name not in ('dU mulq', 'FFN1 gelu', 'FFN1 geluq', 'vocab', 'vocab lse'):        # (the totals stay comparable with earlier rounds' nine products)
            tot[i] += 12 * m
    print('%-14s' % name + ''.join('%12.1f us %5.0f TF' % (m * 1e3, 2.0 * M * N * K / m / 1e9) for m in med))
print('%-14s' % 'x12 per step' + ''.join('%15.2f ms    ' % t for t in tot))
shutil.rmtree(tmp, ignore_errors=True)
