#!/usr/bin/env python3
"""Where the four-wave weight-gradient kernel spends its cycles per K-tile (needs tools/build_alt.sh -DM3P_WG_TL and
M3P_HIP_LIB=m3p_amd/libm3p_hip_alt.so): per-wave s_memtime sums per segment, see the WG_TSEG comment in the template."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import lib as L, ops
lib = L.load()
f = lib.m3p_debug_ring_timeline
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_size_t]
M = int(os.environ.get('AB_M', '41984'))
names = ['phase 1 (64 MFMAs, reads, DMAs)', 'lgkmcnt(0) after phase 1', 'vmcnt(0)', 's_barrier', 'phase 2 (64 MFMAs, reads, DMAs)',
         'lgkmcnt(0) after phase 2', 'step tail / flush / prologue']
for nm, N, K in (('dW lin1', 3072, 768), ('dW lin2', 768, 3072), ('dW qkv', 2304, 768)):
    dy = torch.randn(M, N, device='cuda').to(torch.bfloat16)
    x = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    dw = torch.zeros(N, K, device='cuda')
    for _ in range(3):
        ops.gemm_wgrad(dy, x, dw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm_wgrad(dy, x, dw); e1.record(); torch.cuda.synchronize()
    buf = np.zeros((256, 4, 8), dtype=np.uint64)
    rc = f(buf.ctypes.data, buf.nbytes); assert rc == 0, rc
    d = buf.astype(np.float64)
    live = d[..., 7] > 0
    d = d[live]
    nkt = d[:, 7]
    print('%s M=%d N=%d K=%d: %.1f us (instrumented); %d waves, %.0f K-tiles each' % (nm, M, N, K, e0.elapsed_time(e1) * 1e3, live.sum(), nkt.mean()))
    tot = d[:, :7].sum(1)
    for k, s in enumerate(names):
        per = d[:, k] / nkt
        print('  %-36s %8.0f ticks per K-tile (%4.1f %%)   p10 %.0f  p90 %.0f' % (s, per.mean(), 100 * d[:, k].sum() / tot.sum(), np.percentile(per, 10), np.percentile(per, 90)))
    print('  %-36s %8.0f ticks per K-tile; 128 MFMAs = 2048' % ('all', (tot / nkt).mean()))
