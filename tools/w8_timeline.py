#!/usr/bin/env python3
"""Where the eight-wave 256x256 NT kernel spends its cycles, per epilogue and shape (needs tools/build_alt.sh -DM3P_W8_TL and
M3P_HIP_LIB=m3p_amd/libm3p_hip_alt.so): per-wave s_memtime sums per segment + the spread of the waves' start / end stamps."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import lib as L, ops
lib = L.load()
f = lib.m3p_debug_ring_timeline
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_size_t]
M = int(os.environ.get('AB_M', '41984'))
names = ['K loop', 'bias + aux fetch / wait', 'epilogue pieces (compute, staging, stores)', 'restart (zero, barrier, request, fragments)', 'prologue']
SHAPES = (('FFN1 fwd (bias)', 3072, 768, L.EPI_BIAS), ('FFN1 fwd (bias + GELU + byte)', 3072, 768, L.EPI_BIAS_GELUQ),
          ('dU (byte decode)', 3072, 768, L.EPI_MULQ), ('q/k/v (bias)', 2304, 768, L.EPI_BIAS), ('out_lin (bias+drop+res)', 768, 768, L.EPI_BIAS_DROP_RES))
if os.environ.get('TL_ONLY'):
    SHAPES = tuple(sh for sh in SHAPES if any(k in sh[0] for k in os.environ['TL_ONLY'].split(',')))
lib.m3p_debug_set_variant(int(os.environ.get('M3P_VARIANT', '6')))
for nm, N, K, epi in SHAPES:
    a = torch.randn(M, K, device='cuda').to(torch.bfloat16); w = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
    aux = torch.randn(M, N, device='cuda').to(torch.bfloat16)
    bias = torch.randn(N, device='cuda')
    out = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
    codes = torch.randint(0, 256, (M * N,), dtype=torch.uint8, device='cuda')
    kw = {}
    if epi in (L.EPI_BIAS, L.EPI_BIAS_DROP_RES, L.EPI_BIAS_GELUQ): kw['bias'] = bias
    if epi in (L.EPI_DGELU, L.EPI_BIAS_DROP_RES): kw['aux'] = aux
    if epi == L.EPI_BIAS_DROP_RES: kw.update(seed=5, p_drop=0.1)
    if epi == L.EPI_BIAS_GELUQ: kw['out2'] = codes
    if epi == L.EPI_MULQ: kw.update(aux=codes, colsum=torch.zeros(N, device='cuda'))
    for _ in range(3):
        ops.gemm_nt(a, w, epi, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm_nt(a, w, epi, out=out, **kw); e1.record(); torch.cuda.synchronize()
    buf = np.zeros((2, 256, 8, 8), dtype=np.uint64)
    rc = f(buf.ctypes.data, buf.nbytes); assert rc == 0, rc
    d = buf[0].astype(np.float64)
    kq = buf[1].astype(np.float64)
    live = d[..., 6] > 0                      # workgroups beyond the tile count return before the first stamp
    span = (d[..., 6] - d[..., 5])[live]
    print('%s M=%d N=%d K=%d: %.1f us; %d waves, span mean %.0f ticks (min %.0f max %.0f)'
          % (nm, M, N, K, e0.elapsed_time(e1) * 1e3, live.sum(), span.mean(), span.min(), span.max()))
    d = d[live]
    tiles = (M // 256) * (N // 256)
    for k, s in enumerate(names):
        print('  %-46s %9.0f (%.1f%% of the span; %.0f per tile)' % (s, d[..., k].mean(), 100 * d[..., k].mean() / span.mean(),
                                                                     d[..., k].mean() / (tiles / 256.0)))
    # phases of a K-tile (sums over the wave's K-tiles / their count); each quarter = its memory issue + 16 MFMAs, measured up to
    # the stamp in front of its closing lgkmcnt wait (which therefore counts into the next phase)
    kq = kq[live]
    n_kt = kq[..., 6]
    ph = ['quarter 0', 'quarter 1', 'quarter 2', 'lgkm + vmcnt(0) wait', 's_barrier', 'quarter 3']
    tot = sum(kq[..., i].sum() for i in range(6)) / n_kt.sum()
    print('  K-tile phases, ticks per K-tile (mean over waves; %.0f in all):' % tot)
    for i, nm_ in enumerate(ph):
        per = kq[..., i] / n_kt
        print('    %-22s %7.0f   (waves 0-3 %7.0f, waves 4-7 %7.0f; min %.0f max %.0f)'
              % (nm_, per.mean(), per.reshape(-1, 8)[:, :4].mean() if per.ndim == 1 else per[..., :4].mean(),
                 per.reshape(-1, 8)[:, 4:].mean() if per.ndim == 1 else per[..., 4:].mean(), per.min(), per.max()))
