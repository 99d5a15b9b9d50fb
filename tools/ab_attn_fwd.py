#!/usr/bin/env python3
"""A/B of attention FORWARD builds inside one process (see tools/ab_attn.py): every arm runs m3p_attn_fwd on the same operands at the
benchmarked size (B = 256, S = 164, 12 heads of 64, all keys valid), dropout 0.1 with the keep words written, and dropout off.

    python tools/ab_attn_fwd.py libm3p_hip.so libm3p_hip_r05.so ...        # files under m3p_amd/
"""
import ctypes as C, os, shutil, sys, tempfile
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
    shutil.copy(os.path.join(ROOT, 'm3p_amd', name), path)
    h = C.CDLL(path)
    h.m3p_attn_fwd.restype, h.m3p_attn_fwd.argtypes = L.SIGNATURES['m3p_attn_fwd']
    arms.append((name, h))
st = torch.cuda.current_stream().cuda_stream
M = B * S
qkv = (torch.randn(M, 3 * d, device='cuda') * 0.5).to(torch.bfloat16)
keylen = torch.full((B,), S, dtype=torch.int32, device='cuda')
ctx = torch.empty(M, d, dtype=torch.bfloat16, device='cuda')
lse = torch.empty(B * H * S, dtype=torch.float32, device='cuda')
nt = (S + 15) // 16
keep = torch.zeros(B * H * nt * nt * 4, dtype=torch.int64, device='cuda')
t24, ik = L.thresh24(0.1), 1.0 / 0.9
def run(h, drop, n):
    for _ in range(n):
        rc = h.m3p_attn_fwd(qkv.data_ptr(), keylen.data_ptr(), ctx.data_ptr(), lse.data_ptr(), keep.data_ptr() if drop else None, B, S, H, dh, 5,
                            t24 if drop else 0, ik if drop else 1.0, st)
        assert rc == 0
res = {}
for drop in (True, False):
    times = [[] for _ in arms]
    for _, h in arms: run(h, drop, 3)
    for rnd in range(9):
        for i, (_, h) in enumerate(arms):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(h, drop, 10); e1.record(); torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / 10 * 1e3)
    res[drop] = [(sorted(t)[4], min(t)) for t in times]
for i, (name, _) in enumerate(arms):
    print('%-28s attn_fwd dropout 0.1 + keep words %6.1f us (min %6.1f)   dropout off %6.1f us (min %6.1f)' % ((name,) + res[True][i] + res[False][i]))
shutil.rmtree(tmp, ignore_errors=True)
