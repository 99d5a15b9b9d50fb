#!/usr/bin/env python3
"""Per-shape GEMM table of one cfg2 training step: every (epilogue, M, N, K) instance the step launches, our kernel
next to the vendor GEMM (torch.matmul = hipBLASLt; calibration only, never on the product path) measured in the same
process on the same box, interleaved.  Writes markdown to stdout (committed under profiles/).

    python tools/shape_table.py [B]          # sequences per step, default 256
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p_amd import ops, lib as L   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S, d, V, NP = 164, 768, 250002, 19
M = B * S
n_pred = B * NP
VP = (V + 63) // 64 * 64


def t(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device='cuda') * scale).to(torch.bfloat16)


rows = []


def nt(name, m, n, k, epi, per_step, **kw):
    a, w = rnd(m, k), rnd(n, k, scale=0.05)
    out = torch.empty(m, n, dtype=torch.bfloat16, device='cuda')
    bias = torch.randn(n, device='cuda') if epi in (1, 2, 3) else None
    aux = rnd(m, n) if epi in (3, 4, 5, 6) else None
    cs = torch.zeros(n, device='cuda') if epi == 5 else None
    ours = t(lambda: ops.gemm_nt(a, w, epi, bias=bias, aux=aux, out=out, colsum=cs, p_drop=0.1 if epi == 3 else 0.0, seed=3, **kw))
    blas = t(lambda: torch.matmul(a, w.t(), out=out))
    rows.append((name, 'nt/%s' % ['none', 'bias', 'bias_gelu', 'bias_drop_res', 'res', 'dgelu', 'mul'][epi], m, n, k, per_step, ours, blas))


def wg(name, m, n, k, per_step):
    dy, x = rnd(m, n, scale=0.1), rnd(m, k)
    dw = torch.zeros(n, k, device='cuda')
    dwb = torch.empty(n, k, dtype=torch.bfloat16, device='cuda')
    ours = t(lambda: ops.gemm_wgrad(dy, x, dw))
    blas = t(lambda: torch.matmul(dy.t(), x, out=dwb))
    rows.append((name, 'wgrad', m, n, k, per_step, ours, blas))


nt('QKV fwd', M, 3 * d, d, 1, 12, scale_cols=d, scale=0.125)
nt('out_lin fwd (+dropout +residual)', M, d, d, 3, 12)
nt('FFN lin1 fwd', M, 4 * d, d, 1, 12)
nt('FFN lin2 fwd (+dropout +residual)', M, d, 4 * d, 3, 12)
nt('dU = dY2 W2 * gelu\'(u) (+bias colsum)', M, 4 * d, d, 5, 12)
nt('dx1 = dU W1 + res', M, d, 4 * d, 4, 12)
nt('dctx = dAO Wo', M, d, d, 0, 12)
nt('dh = dqkv Wqkv + res', M, d, 3 * d, 4, 12)
nt('region projection', B * 36, d, 2048, 1, 1)
wg('dW lin2', M, d, 4 * d, 12)
wg('dW lin1', M, 4 * d, d, 12)
wg('dW out_lin', M, d, d, 12)
wg('dW qkv', M, 3 * d, d, 12)
wg('dW region projection', B * 36, d, 2048, 1)
# vocabulary block
hsel, E = rnd(n_pred, d), rnd(V, d, scale=0.05)
logits = torch.empty(n_pred, VP, dtype=torch.bfloat16, device='cuda')
bias = torch.randn(V, device='cuda')
ours = t(lambda: ops.gemm_nt(hsel, E, 1, bias=bias, out=logits, n=V), 5)
blas = t(lambda: torch.matmul(hsel, E.t(), out=logits[:, :V]), 5)
rows.append(('vocabulary projection', 'nt/bias', n_pred, V, d, 1, ours, blas))
dlog = rnd(n_pred, VP, scale=0.01)
dlog[:, V:] = 0
dH = torch.zeros(n_pred, d, device='cuda')
dHb = torch.empty(n_pred, d, dtype=torch.bfloat16, device='cuda')
ours = t(lambda: ops.gemm_nn_streamk(dlog, E, dH), 5)
blas = t(lambda: torch.matmul(dlog[:, :V], E, out=dHb), 5)
rows.append(('vocabulary dgrad (stream-K, fp32 atomics)', 'nn/streamk', n_pred, d, V, 1, ours, blas))
dE = torch.zeros(V, d, device='cuda')
dEb = torch.empty(V, d, dtype=torch.bfloat16, device='cuda')
ours = t(lambda: ops.gemm_wgrad(dlog, hsel, dE, n=V, k=d), 5)
blas = t(lambda: torch.matmul(dlog[:, :V].t(), hsel, out=dEb), 5)
rows.append(('vocabulary wgrad', 'wgrad', n_pred, V, d, 1, ours, blas))

print('| GEMM | kind | M | N | K | launches / step | ours us | ours TF/s | vendor us | vendor TF/s | ours / vendor | ms / step (ours) |')
print('|---|---|---|---|---|---|---|---|---|---|---|---|')
tot_o = tot_b = 0.0
for name, kind, m, n, k, per, o, b in rows:
    f = 2.0 * m * n * k / 1e9
    tot_o += per * o
    tot_b += per * b
    print('| %s | %s | %d | %d | %d | %d | %.1f | %.0f | %.1f | %.0f | %.2f | %.2f |'
          % (name, kind, m, n, k, per, o * 1e3, f / o, b * 1e3, f / b, b / o, per * o))
print('\nGEMM time per step: ours %.2f ms, vendor kernels (plain products, no epilogues) %.2f ms' % (tot_o, tot_b))
