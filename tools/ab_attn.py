#!/usr/bin/env python3
"""A/B of attention builds inside ONE process (see tools/ab_gemm.py): m3p_attn_fwd once (for lse and the keep-bit words), then
m3p_attn_bwd of every arm on the same operands at the benchmarked size (B = 256, S = 164, 12 heads of 64, dropout 0.1);
dq/dk/dv must equal the first arm's bits (the bias sums only to rounding: their summation order changes).

    python tools/ab_attn.py libm3p_hip.so libm3p_hip_attn14.so ...        # files under m3p_amd/
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

B, S, H, dh = int(os.environ.get('AB_B', '256')), int(os.environ.get('AB_S', '164')), 12, 64
d = H * dh
tmp = tempfile.mkdtemp()
arms = []
for k, name in enumerate(sys.argv[1:]):
    path = os.path.join(tmp, 'arm%d.so' % k)
    shutil.copy(os.path.join(ROOT, 'm3p_amd', name.split(':')[0]), path)      # <file>[:<m3p_debug_attn_variant value>]
    h = C.CDLL(path)
    if ':' in name:
        h.m3p_debug_attn_variant(int(name.split(':')[1]))
    for fn in ('m3p_attn_fwd', 'm3p_attn_bwd'):
        getattr(h, fn).restype, getattr(h, fn).argtypes = L.SIGNATURES[fn]
    arms.append((name, h))
st = torch.cuda.current_stream().cuda_stream
BF = torch.bfloat16
M = B * S
qkv = (torch.randn(M, 3 * d, device='cuda') * 0.5).to(BF)
dctx = torch.randn(M, d, device='cuda').to(BF)
keylen = torch.full((B,), S, dtype=torch.int32, device='cuda')
keylen[::3] = S - 37
ctx = torch.empty(M, d, dtype=BF, device='cuda')
lse = torch.empty(B * H * S, dtype=torch.float32, device='cuda')
nt = (S + 15) // 16
keep = torch.zeros(B * H * nt * nt * 4, dtype=torch.int64, device='cuda')
t24, ik = L.thresh24(0.1), 1.0 / 0.9
rc = arms[0][1].m3p_attn_fwd(qkv.data_ptr(), keylen.data_ptr(), ctx.data_ptr(), lse.data_ptr(), keep.data_ptr(), B, S, H, dh, 5, t24, ik, st)
assert rc == 0
outs = []
for name, h in arms:
    dqkv = torch.zeros(M, 3 * d, dtype=BF, device='cuda')
    dbias = torch.zeros(3 * d, device='cuda')

    def bwd(h=h, dqkv=dqkv, dbias=dbias):
        rc = h.m3p_attn_bwd(qkv.data_ptr(), keylen.data_ptr(), ctx.data_ptr(), dctx.data_ptr(), lse.data_ptr(), keep.data_ptr(),
                            dqkv.data_ptr(), dbias.data_ptr(), B, S, H, dh, 0.125, 5, t24, ik, st)
        assert rc == 0
    bwd()
    torch.cuda.synchronize()
    outs.append((bwd, dqkv.clone(), dbias.clone()))
for i in range(1, len(arms)):
    print('%s vs %s: dqkv identical %s, bias rel diff %.1e' % (arms[i][0], arms[0][0], torch.equal(outs[i][1], outs[0][1]),
                                                               float((outs[i][2] - outs[0][2]).norm() / outs[0][2].norm())))
times = [[] for _ in arms]
for rnd in range(9):
    for i, (bwd, _, _) in enumerate(outs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            bwd()
        e1.record()
        torch.cuda.synchronize()
        times[i].append(e0.elapsed_time(e1) / 10 * 1e3)
for i, (name, _) in enumerate(arms):
    print('%-28s attn_bwd %6.1f us (min %6.1f)' % (name, sorted(times[i])[4], min(times[i])))
shutil.rmtree(tmp, ignore_errors=True)
