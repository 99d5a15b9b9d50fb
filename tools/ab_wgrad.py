#!/usr/bin/env python3
"""A/B of weight-gradient kernel builds inside ONE process (see tools/ab_gemm.py): every library is a separate CDLL handle, the
arms alternate on the same operands, the median of several rounds is reported; with --check the results must agree with the
first arm's to fp32 rounding.      python tools/ab_wgrad.py libm3p_hip.so libm3p_hip_alt.so [--check]"""
import ctypes as C
import os
import shutil
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m3p_amd import lib as L   # noqa: E402

check = '--check' in sys.argv
names = [a for a in sys.argv[1:] if not a.startswith('--')]      # <file under m3p_amd/>[:<m3p_debug_set_variant value>]
tmp = tempfile.mkdtemp()
arms = []
for k, name in enumerate(names):
    path = os.path.join(tmp, 'arm%d.so' % k)
    shutil.copy(os.path.join(ROOT, 'm3p_amd', name.split(':')[0]), path)
    h = C.CDLL(path)
    if ':' in name:
        h.m3p_debug_set_variant(int(name.split(':')[1]))
    for fn in ('m3p_gemm_wgrad_bf16', 'm3p_gemm_wgrad_workspace_bytes'):
        getattr(h, fn).restype, getattr(h, fn).argtypes = L.SIGNATURES[fn]
    arms.append((name, h))
st = torch.cuda.current_stream().cuda_stream
M = int(os.environ.get('AB_M', '41984'))
SHAPES = [('dW lin2', 768, 3072), ('dW lin1', 3072, 768), ('dW qkv', 2304, 768), ('dW out_lin', 768, 768)]
# (one zero-filled workspace per arm: since round 5 the kernels keep per-tile counters behind the slots, where earlier builds wrote tile ids)
wss = [torch.zeros(a[1].m3p_gemm_wgrad_workspace_bytes(), dtype=torch.uint8, device='cuda') for a in arms]
print('%-12s' % 'shape' + ''.join('%26s' % a[0] for a in arms))
tot = [0.0] * len(arms)
for name, N, K in SHAPES:
    dy = torch.randn(M, N, device='cuda').to(torch.bfloat16)
    x = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    outs = [torch.zeros(N, K, device='cuda') for _ in arms]

    def run(i, n):
        for _ in range(n):
            rc = arms[i][1].m3p_gemm_wgrad_bf16(dy.data_ptr(), N, x.data_ptr(), K, outs[i].data_ptr(), K, M, N, K, 1.0, wss[i].data_ptr(), wss[i].numel(), st)
            assert rc == 0, rc
    for i in range(len(arms)):
        run(i, 1)
    torch.cuda.synchronize()
    if check:
        for i in range(1, len(arms)):
            err = float((outs[i] - outs[0]).abs().max() / outs[0].abs().max())
            assert err < 1e-5, (name, arms[i][0], err)
    times = [[] for _ in arms]
    for rnd in range(7):
        for i in range(len(arms)):
            run(i, 2)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(i, 10)
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / 10)
    med = [sorted(t)[len(t) // 2] for t in times]
    for i, m in enumerate(med):
        tot[i] += 12 * m
    print('%-12s' % name + ''.join('%14.1f us %5.0f TF' % (m * 1e3, 2.0 * M * N * K / m / 1e9) for m in med))
print('%-12s' % 'x12 per step' + ''.join('%19.2f ms    ' % t for t in tot))
