#!/usr/bin/env python3
"""A/B of LayerNorm builds inside ONE process (see tools/ab_gemm.py): forward with the layer-end mask, backward with the
dropout-masked second output and the bias column sums - the two launches of every encoder layer - at the benchmarked
size; also checks that every arm returns the first arm's bits.

    python tools/ab_ln.py libm3p_hip.so libm3p_hip_lnbase.so ...        # files under m3p_amd/
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

rows, d = int(os.environ.get('AB_M', '41984')), int(os.environ.get('AB_D', '768'))
tmp = tempfile.mkdtemp()
arms = []
for k, name in enumerate(sys.argv[1:]):
    path = os.path.join(tmp, 'arm%d.so' % k)
    shutil.copy(os.path.join(ROOT, 'm3p_amd', name), path)
    h = C.CDLL(path)
    for fn in ('m3p_layernorm_fwd', 'm3p_layernorm_bwd'):
        getattr(h, fn).restype, getattr(h, fn).argtypes = L.SIGNATURES[fn]
    arms.append((name, h))
st = torch.cuda.current_stream().cuda_stream
BF = torch.bfloat16
x = torch.randn(rows, d, device='cuda').to(BF)
dy = torch.randn(rows, d, device='cuda').to(BF)
gamma, beta = torch.randn(d, device='cuda'), torch.randn(d, device='cuda')
mask = (torch.rand(rows, device='cuda') < 0.9).to(torch.uint8)
outs = []
for name, h in arms:
    y = torch.empty_like(x); mean = torch.empty(rows, device='cuda'); rstd = torch.empty(rows, device='cuda')
    dx = torch.empty_like(x); dxd = torch.empty_like(x)
    dg, db, dbd = torch.zeros(d, device='cuda'), torch.zeros(d, device='cuda'), torch.zeros(d, device='cuda')

    def fwd():
        rc = h.m3p_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mask.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                 rstd.data_ptr(), rows, d, 1e-12, st)
        assert rc == 0

    def bwd():
        rc = h.m3p_layernorm_bwd(dy.data_ptr(), None, x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), mask.data_ptr(),
                                 dx.data_ptr(), dxd.data_ptr(), dg.data_ptr(), db.data_ptr(), dbd.data_ptr(), rows, d, 77,
                                 L.thresh24(0.1), 1.0 / 0.9, st)
        assert rc == 0
    fwd(); bwd()
    torch.cuda.synchronize()
    outs.append((fwd, bwd, [t.clone() for t in (y, mean, rstd, dx, dxd)], dg.clone()))
for i in range(1, len(arms)):
    same = all(torch.equal(a, b) for a, b in zip(outs[0][2], outs[i][2]))
    print('%s == %s: %s (dgamma rel diff %.1e)' % (arms[i][0], arms[0][0], same,
                                                   float((outs[i][3] - outs[0][3]).norm() / outs[0][3].norm())))
times = [[[], []] for _ in arms]
for rnd in range(9):
    for i, (fwd, bwd, _, _) in enumerate(outs):
        for j, fn in enumerate((fwd, bwd)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[i][j].append(e0.elapsed_time(e1) / 20 * 1e3)
for i, (name, _) in enumerate(arms):
    f, b = sorted(times[i][0])[4], sorted(times[i][1])[4]
    print('%-28s fwd %6.1f us (%4.2f TB/s)   bwd %6.1f us (%4.2f TB/s)' % (name, f, 2 * rows * d * 2 / f / 1e6, b, 4 * rows * d * 2 / b / 1e6))
shutil.rmtree(tmp, ignore_errors=True)
