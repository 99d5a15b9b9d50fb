#!/usr/bin/env python3
"""A/B of fused-Adam builds inside ONE process: m3p_adam_step_ranges over the cfg2 arena (280 M elements as the step's two pieces -
the 192 M-element vocabulary range without the gradient zeroing, the rest with it), arms alternate, median of 9 rounds.
    python tools/ab_adam.py libm3p_hip.so libm3p_hip_adamq4.so ...        # files under m3p_amd/ (tools/build_variant.sh)"""
import ctypes as C, os, shutil, sys, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m3p_amd import lib as L   # noqa: E402
tmp = tempfile.mkdtemp()
arms = []
for k, name in enumerate(sys.argv[1:]):
    path = os.path.join(tmp, 'arm%d.so' % k)
    shutil.copy(os.path.join(ROOT, 'm3p_amd', name), path)
    h = C.CDLL(path)
    h.m3p_adam_step_ranges.restype, h.m3p_adam_step_ranges.argtypes = L.SIGNATURES['m3p_adam_step_ranges']
    arms.append((name, h))
n, nv = 279_937_024, 192_086_016
p, g, m, v = (torch.randn(n, device='cuda') * 0.01 for _ in range(4))
v.abs_()
w16 = torch.zeros(n, dtype=torch.bfloat16, device='cuda')
gn = torch.full((1,), 25.0, dtype=torch.float64, device='cuda')
st = torch.cuda.current_stream().cuda_stream
starts = (C.c_longlong * 2)(0, nv); counts = (C.c_longlong * 2)(nv, n - nv)
steps = (C.c_float * 2)(1e-4, 1e-4); zeros = (C.c_int * 2)(0, 1)


def run(h, k):
    for _ in range(k):
        rc = h.m3p_adam_step_ranges(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), w16.data_ptr(), starts, counts, steps, zeros, 2,
                                    1e-4, 0.9, 0.98, 1e-8, 0.0, gn.data_ptr(), 5.0, 1.0, st)
        assert rc == 0, rc


times = [[] for _ in arms]
for rnd in range(9):
    for i, (_, h) in enumerate(arms):
        run(h, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(h, 5); e1.record(); torch.cuda.synchronize()
        times[i].append(e0.elapsed_time(e1) / 5)
gb = (nv * 30 + (n - nv) * 34) / 1e9
for (name, _), t in zip(arms, times):
    med = sorted(t)[len(t) // 2]
    print('%-32s %7.1f us  %.2f TB/s' % (name, med * 1e3, gb / med))
