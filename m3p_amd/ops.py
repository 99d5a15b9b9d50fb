"""Raw (non-autograd) launchers: one Python function per C-ABI entry point.

Each allocates its outputs with torch (device memory + caching allocator are plumbing),
passes raw device pointers and the current HIP stream across the C ABI and raises on any
error code.  The autograd layer (m3p_amd/functional.py) composes these."""
import ctypes as C
import os

import torch

from . import lib as L

BF16 = torch.bfloat16
_DEBUG_CHECKS = os.environ.get('M3P_DEBUG_CHECKS') == '1'      # host-synchronising consistency checks (tests, debugging)

# bench.py sets this to a dict to time individual GEMM launches with HIP events recorded on
# the launch stream: {(kind, M, N, K): [(start_event, end_event), ...]}
PROFILE = None
_EPI_NAMES = ['gemm_nt/none', 'gemm_nt/bias', 'gemm_nt/bias_gelu', 'gemm_nt/bias_drop_res', 'gemm_nt/res',
              'gemm_nt/dgelu', 'gemm_nt/mul', 'gemm_nt/mulq', 'gemm_nt/bias_geluq', 'gemm_nt/bias_lse']


PROFILE_ONLY = None     # if set: the one (kind, M, N, K) instance that is timed (bench.py: the dominant kernel)


def _prof_begin(key=None):
    if PROFILE is None or (PROFILE_ONLY is not None and key != PROFILE_ONLY):
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _prof_end(e0, key):
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        PROFILE.setdefault(key, []).append((e0, e1))


def _chk_bf16(*ts):
    for t in ts:
        if t is not None:
            assert t.dtype == BF16 and t.is_cuda, (t.dtype, t.device)


def gemm_nt(a, w, epilogue=L.EPI_NONE, bias=None, aux=None, out=None, out2=None, colsum=None,
            scale_cols=0, scale=1.0, alpha=1.0, seed=0, p_drop=0.0, n=None, out8=None, scale8=None, amax8=None, out8_bf8=False):
    """C[M,N] = epi(a[M,K] @ w[N,K]^T).  a, w bf16 (row pitch = stride(0)); returns C (bf16).
    ``n`` restricts the number of output columns (rows of w) used.
    out8 (EPI_BIAS_GELUQ / EPI_MULQ): uint8 [M, N] that receives the 8-bit copy of C - sat(C * scale8) in e4m3, or e5m2 with
    out8_bf8 - for the fp8 product that consumes C; amax8 (fp32 [1], zeroed by the caller) is raised to max |C|."""
    M, K = a.shape
    N = w.shape[0] if n is None else n
    if epilogue == L.EPI_MULQ:     # aux = the byte codes of gelu_fwd_gq for this [M, N], in the GEMM's fragment order
        _chk_bf16(a, w, out)
        assert aux is not None and aux.dtype == torch.uint8 and aux.is_contiguous() and aux.numel() == M * N
    elif epilogue == L.EPI_BIAS_LSE:        # out2 = float32 [N / 64, M, 2] block statistics; scale_cols = V (valid columns)
        _chk_bf16(a, w, out)
        assert out2 is not None and out2.dtype == torch.float32 and out2.is_contiguous() and out2.numel() == (N // 64) * M * 2
        assert bias is not None and 0 < scale_cols <= N
    elif epilogue == L.EPI_BIAS_GELUQ:      # out2 = where those codes go (uint8 [M * N]); C = gelu(a w^T + bias)
        _chk_bf16(a, w, out)
        assert out2 is not None and out2.dtype == torch.uint8 and out2.is_contiguous() and out2.numel() == M * N and bias is not None
    else:
        _chk_bf16(a, w, aux, out, out2)
    assert w.shape[1] == K and a.stride(1) == 1 and w.stride(1) == 1
    if out is None:
        out = torch.empty((M, N), dtype=BF16, device=a.device)
    ep = L.Epilogue()
    ep.bias = L.ptr(bias)
    ep.aux = L.ptr(aux)
    ep.out2 = L.ptr(out2)
    ep.colsum = L.ptr(colsum)
    ep.ld_aux = aux.stride(0) if (aux is not None and epilogue != L.EPI_MULQ) else 0
    ep.ld_out2 = out2.stride(0) if (out2 is not None and epilogue not in (L.EPI_BIAS_GELUQ, L.EPI_BIAS_LSE)) else 0
    ep.scale_cols = scale_cols
    if epilogue == L.EPI_BIAS_LSE:
        ep.ld_out2, ep.scale_cols = scale_cols, 0
    ep.scale = scale
    ep.alpha = alpha
    ep.seed = seed
    ep.thresh24 = L.thresh24(p_drop)
    ep.inv_keep = 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0
    if bias is not None:
        assert bias.dtype == torch.float32
    if out8 is not None:
        assert epilogue in (L.EPI_BIAS_GELUQ, L.EPI_MULQ) and out8.dtype == torch.uint8 and out8.shape == (M, N) and out8.stride(1) == 1
        ep.out8, ep.ld_out8, ep.out8_bf8 = out8.data_ptr(), out8.stride(0), 1 if out8_bf8 else 0
        ep.scale8, ep.amax8 = L.ptr(scale8), L.ptr(amax8)
    e0 = _prof_begin((_EPI_NAMES[epilogue], M, N, K))
    rc = L.load().m3p_gemm_nt_bf16(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(),
                                   out.stride(0), M, N, K, epilogue, C.byref(ep), L.stream())
    L.check(rc, 'm3p_gemm_nt_bf16')
    _prof_end(e0, (_EPI_NAMES[epilogue], M, N, K))
    return out


def quant_fp8(x, scale=None, amax=None, bf8=False, out=None):
    """x bf16 [rows, cols] -> 8-bit [rows, cols] (uint8 storage): fp8 e4m3, or bf8 e5m2 when ``bf8``; multiplied by the
    device scalar ``scale`` first, saturating; ``amax`` (device fp32 scalar, zeroed by the caller) is raised to max |x|."""
    _chk_bf16(x)
    rows, cols = x.shape
    assert x.stride(1) == 1
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.uint8, device=x.device)
    rc = L.load().m3p_quant_fp8(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), rows, cols, L.ptr(scale), L.ptr(amax),
                                1 if bf8 else 0, L.stream())
    L.check(rc, 'm3p_quant_fp8')
    return out


def quant_fp8_batch(desc, n_desc, blocks_per_matrix=512):
    """Many contiguous bf16 matrices -> e4m3 in one launch; desc int64 [n_desc, 5] on the device (csrc/optim.hip)."""
    L.check(L.load().m3p_quant_fp8_batch(desc.data_ptr(), n_desc, blocks_per_matrix, L.stream()), 'm3p_quant_fp8_batch')


def gemm_nt_fp8(a8, w8, epilogue=L.EPI_NONE, a_is_bf8=False, descale_a=None, descale_b=None, bias=None, aux=None, out=None,
                colsum=None, scale_cols=0, scale=1.0, seed=0, p_drop=0.0):
    """C[M,N] (bf16) = epi(descale_a * descale_b * a8[M,K] @ w8[N,K]^T): 8-bit operands (uint8 storage) from quant_fp8,
    descale_* device fp32 scalars (1 / the quantisation scales)."""
    assert a8.dtype == torch.uint8 and w8.dtype == torch.uint8 and a8.is_cuda and w8.is_cuda
    M, K = a8.shape
    N = w8.shape[0]
    assert w8.shape[1] == K and a8.stride(1) == 1 and w8.stride(1) == 1
    _chk_bf16(aux, out)
    if out is None:
        out = torch.empty((M, N), dtype=BF16, device=a8.device)
    ep = L.Epilogue()
    ep.bias = L.ptr(bias)
    ep.aux = L.ptr(aux)
    ep.colsum = L.ptr(colsum)
    ep.ld_aux = aux.stride(0) if aux is not None else 0
    ep.scale_cols = scale_cols
    ep.scale = scale
    ep.alpha = 1.0
    ep.seed = seed
    ep.thresh24 = L.thresh24(p_drop)
    ep.inv_keep = 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0
    ep.descale_a = L.ptr(descale_a)
    ep.descale_b = L.ptr(descale_b)
    e0 = _prof_begin(('gemm_fp8/' + _EPI_NAMES[epilogue].split('/')[1], M, N, K))
    rc = L.load().m3p_gemm_nt_fp8(a8.data_ptr(), a8.stride(0), 1 if a_is_bf8 else 0, w8.data_ptr(), w8.stride(0), out.data_ptr(),
                                  out.stride(0), M, N, K, epilogue, C.byref(ep), L.stream())
    L.check(rc, 'm3p_gemm_nt_fp8')
    _prof_end(e0, ('gemm_fp8/' + _EPI_NAMES[epilogue].split('/')[1], M, N, K))
    return out


def gemm_nt_streamk(a, w, out_f32, alpha=1.0):
    """out_f32[M,N] (fp32) += alpha * a[M,K] @ w[N,K]^T  (stream-K, fp32 atomics)."""
    _chk_bf16(a, w)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and out_f32.dtype == torch.float32 and out_f32.shape == (M, N)
    e0 = _prof_begin(('gemm_nt/streamk', M, N, K))
    rc = L.load().m3p_gemm_nt_streamk_f32(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out_f32.data_ptr(),
                                          out_f32.stride(0), M, N, K, alpha, L.stream())
    L.check(rc, 'm3p_gemm_nt_streamk_f32')
    _prof_end(e0, ('gemm_nt/streamk', M, N, K))
    return out_f32


def gemm_nn_streamk(a, w_kn, out_f32, alpha=1.0):
    """out_f32[M,N] (fp32) += alpha * a[M,K] @ w_kn[:K]  with w_kn [k_valid <= K, N] row-major (stream-K, fp32
    atomics); columns k_valid..K of ``a`` must be zeros."""
    _chk_bf16(a, w_kn)
    M, K = a.shape
    k_valid, N = w_kn.shape
    assert k_valid <= K and out_f32.dtype == torch.float32 and out_f32.shape == (M, N) and w_kn.stride(1) == 1
    e0 = _prof_begin(('gemm_nt/streamk', M, N, K))
    rc = L.load().m3p_gemm_nn_streamk_f32(a.data_ptr(), a.stride(0), w_kn.data_ptr(), w_kn.stride(0), k_valid,
                                          out_f32.data_ptr(), out_f32.stride(0), M, N, K, alpha, L.stream())
    L.check(rc, 'm3p_gemm_nn_streamk_f32')
    _prof_end(e0, ('gemm_nt/streamk', M, N, K))
    return out_f32


def gemm_nn(a, w_kn, out_f32, alpha=1.0, k_rows_readable=None):
    """out_f32[M,N] (fp32) += alpha * a[M,K] @ w_kn[:K].  Whole-tile shapes whose second operand is readable (finite) for all
    K rows - k_rows_readable >= K: the rows behind w_kn's own k_valid are memory the caller vouches for, e.g. the arena
    behind the vocabulary matrix - run on the four-wave kernel (m3p_gemm_nn_w4_f32, no atomics); everything else on the
    stream-K form."""
    _chk_bf16(a, w_kn)
    M, K = a.shape
    k_valid, N = w_kn.shape
    if (k_rows_readable or k_valid) >= K and M % 256 == 0 and N % 256 == 0 and K % 64 == 0 and K >= 4096:
        assert out_f32.dtype == torch.float32 and out_f32.shape == (M, N) and w_kn.stride(1) == 1
        e0 = _prof_begin(('gemm_nn/w4', M, N, K))
        ws = _wgrad_workspace(a.device)
        rc = L.load().m3p_gemm_nn_w4_f32(a.data_ptr(), a.stride(0), w_kn.data_ptr(), w_kn.stride(0), out_f32.data_ptr(),
                                         out_f32.stride(0), M, N, K, alpha, ws.data_ptr(), ws.numel(), L.stream())
        if rc == 0:
            _prof_end(e0, ('gemm_nn/w4', M, N, K))
            return out_f32
        if rc != -2:
            L.check(rc, 'm3p_gemm_nn_w4_f32')
    return gemm_nn_streamk(a, w_kn, out_f32, alpha)


_WGRAD_WS = {}     # (device index, stream) -> workspace tensor of the four-wave weight-gradient kernel


def _wgrad_workspace(device):
    """One scratch buffer per (device, stream): launches on one stream are ordered, so they can share it."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _WGRAD_WS.get(key)
    if ws is None:
        ws = _WGRAD_WS[key] = torch.empty(int(L.load().m3p_gemm_wgrad_workspace_bytes()), dtype=torch.uint8, device=device)
    return ws


def gemm_wgrad(dy, x, dw, alpha=1.0, n=None, k=None, dw_is_zero=False, report_store=False, must_store=False):
    """dw[N,K] (fp32) += alpha * dy[M,N]^T @ x[M,K].  dw_is_zero: the caller knows dw[:N, :K] to hold zeros - shapes dealt out
    as whole tiles (the vocabulary matrix) are then stored instead of accumulated with atomics; same result.  must_store: dw is
    only LOGICALLY zero (the arena's lazily zeroed range): a launcher that declines the store raises here instead of
    accumulating onto what the buffer physically holds."""
    _chk_bf16(dy, x)
    assert dw.dtype == torch.float32 and dw.stride(-1) == 1
    M = dy.shape[0]
    N = dy.shape[1] if n is None else n
    K = x.shape[1] if k is None else k
    assert x.shape[0] == M and dw.shape[0] >= N and dw.shape[1] >= K
    e0 = _prof_begin(('gemm_wgrad', M, N, K))
    ws = _wgrad_workspace(dy.device)
    if dw_is_zero:
        rc = L.load().m3p_gemm_wgrad_store_bf16(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dw.data_ptr(),
                                                dw.stride(0), M, N, K, alpha, ws.data_ptr(), ws.numel(), L.stream())
        if rc == 0:
            _prof_end(e0, ('gemm_wgrad', M, N, K))
            return True if report_store else dw
        if rc != -2:        # (M3P_ENOTIMPL: not a whole-tile shape - accumulate below)
            L.check(rc, 'm3p_gemm_wgrad_store_bf16')
        if must_store:
            raise L.M3PError('m3p_gemm_wgrad_store_bf16 declined (M3P_ENOTIMPL) a store the caller depends on: dw is not physically zero')
    rc = L.load().m3p_gemm_wgrad_bf16(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dw.data_ptr(),
                                      dw.stride(0), M, N, K, alpha, ws.data_ptr(), ws.numel(), L.stream())
    L.check(rc, 'm3p_gemm_wgrad_bf16')
    _prof_end(e0, ('gemm_wgrad', M, N, K))
    return False if report_store else dw


def gemm_wgrad_pair(dy_a, x_a, dw_a, dy_b, x_b, dw_b, alpha=1.0):
    """dw_a += alpha * dy_a^T @ x_a and dw_b += alpha * dy_b^T @ x_b (same M) in one launch where the shapes allow."""
    _chk_bf16(dy_a, x_a, dy_b, x_b)
    assert dw_a.dtype == torch.float32 and dw_b.dtype == torch.float32 and dw_a.stride(-1) == 1 and dw_b.stride(-1) == 1
    M = dy_a.shape[0]
    assert x_a.shape[0] == M and dy_b.shape[0] == M and x_b.shape[0] == M
    Na, Ka, Nb, Kb = dy_a.shape[1], x_a.shape[1], dy_b.shape[1], x_b.shape[1]
    assert dw_a.shape[0] >= Na and dw_a.shape[1] >= Ka and dw_b.shape[0] >= Nb and dw_b.shape[1] >= Kb
    e0 = _prof_begin(('gemm_wgrad_pair', M, Na + Nb, Ka))
    ws = _wgrad_workspace(dy_a.device)
    rc = L.load().m3p_gemm_wgrad_pair_bf16(dy_a.data_ptr(), dy_a.stride(0), x_a.data_ptr(), x_a.stride(0), dw_a.data_ptr(), dw_a.stride(0),
                                           Na, Ka, dy_b.data_ptr(), dy_b.stride(0), x_b.data_ptr(), x_b.stride(0), dw_b.data_ptr(),
                                           dw_b.stride(0), Nb, Kb, M, alpha, ws.data_ptr(), ws.numel(), L.stream())
    L.check(rc, 'm3p_gemm_wgrad_pair_bf16')
    _prof_end(e0, ('gemm_wgrad_pair', M, Na + Nb, Ka))


def layernorm_fwd(x, gamma, beta, rowmask=None, eps=1e-12):
    _chk_bf16(x)
    rows, d = x.shape
    assert x.is_contiguous()
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    rc = L.load().m3p_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), L.ptr(rowmask), y.data_ptr(),
                                    mean.data_ptr(), rstd.data_ptr(), rows, d, eps, L.stream())
    L.check(rc, 'm3p_layernorm_fwd')
    return y, mean, rstd


def layernorm_bwd(dy_a, dy_b, x, gamma, mean, rstd, rowmask, dgamma, dbeta, dbias_drop=None,
                  want_drop=False, seed=0, p_drop=0.0):
    """Returns (dx, dx_drop).  dx_drop is dx pushed through the residual-branch dropout
    (None unless want_drop).  dgamma/dbeta/dbias_drop are accumulated in place (fp32)."""
    _chk_bf16(dy_a, dy_b, x)
    rows, d = x.shape
    dx = torch.empty_like(x)
    dx_drop = torch.empty_like(x) if want_drop else None
    rc = L.load().m3p_layernorm_bwd(dy_a.data_ptr(), L.ptr(dy_b), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(),
                                    rstd.data_ptr(), L.ptr(rowmask), dx.data_ptr(), L.ptr(dx_drop),
                                    dgamma.data_ptr(), dbeta.data_ptr(), L.ptr(dbias_drop), rows, d, seed,
                                    L.thresh24(p_drop), 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0, L.stream())
    L.check(rc, 'm3p_layernorm_bwd')
    return dx, dx_drop


def attn_fwd(qkv, keylen, B, S, H, dh, seed=0, p_drop=0.0, want_mask=False):
    """qkv bf16 [B*S, 3*H*dh] -> (ctx bf16 [B*S, H*dh], lse fp32 [B,H,S]) and, with want_mask, the dropout
    keep-bit words for attn_bwd (None when p_drop == 0)."""
    _chk_bf16(qkv)
    assert qkv.is_contiguous() and qkv.shape == (B * S, 3 * H * dh) and keylen.dtype == torch.int32
    ctx = torch.empty((B * S, H * dh), dtype=BF16, device=qkv.device)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=qkv.device)
    mask = None
    if want_mask and p_drop > 0:
        nt = (S + 15) // 16
        mask = torch.empty((B * H, nt, nt, 4), dtype=torch.int64, device=qkv.device)
    rc = L.load().m3p_attn_fwd(qkv.data_ptr(), keylen.data_ptr(), ctx.data_ptr(), lse.data_ptr(), L.ptr(mask), B, S, H, dh,
                               seed, L.thresh24(p_drop), 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0, L.stream())
    L.check(rc, 'm3p_attn_fwd')
    return (ctx, lse, mask) if want_mask else (ctx, lse)


def attn_bwd(qkv, keylen, ctx, dctx, lse, B, S, H, dh, dbias_qkv=None, seed=0, p_drop=0.0, keepmask=None):
    """dqkv of attn_fwd.  With dropout on, keepmask = the words attn_fwd(..., want_mask=True) returned (required)."""
    _chk_bf16(qkv, ctx, dctx)
    assert dctx.is_contiguous() and ctx.is_contiguous()
    assert p_drop == 0 or keepmask is not None, 'attention backward with dropout needs the keep words of the forward pass'
    dqkv = torch.empty_like(qkv)
    rc = L.load().m3p_attn_bwd(qkv.data_ptr(), keylen.data_ptr(), ctx.data_ptr(), dctx.data_ptr(), lse.data_ptr(),
                               L.ptr(keepmask), dqkv.data_ptr(), L.ptr(dbias_qkv), B, S, H, dh, 1.0 / (dh ** 0.5), seed,
                               L.thresh24(p_drop), 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0, L.stream())
    L.check(rc, 'm3p_attn_bwd')
    return dqkv


def attn_query_fwd(q, kv, klen, B, Tq, H, dh, Lk, causal=False, pos0=0):
    """Decoder-inference attention: q bf16 [B*Tq, >= H*dh] (scaled), kv bf16 [B, >= Lk, >= 2*H*dh] (keys | values per
    cached position), klen int32 [B] or None -> ctx bf16 [B*Tq, H*dh]  (csrc/decode.hip)."""
    _chk_bf16(q, kv)
    assert q.stride(1) == 1 and kv.dim() == 3 and kv.stride(2) == 1 and kv.shape[0] == B and kv.shape[1] >= Lk
    ctx = torch.empty((B * Tq, H * dh), dtype=BF16, device=q.device)
    rc = L.load().m3p_attn_query_fwd(q.data_ptr(), q.stride(0), kv.data_ptr(), kv.stride(0), kv.stride(1), L.ptr(klen),
                                     ctx.data_ptr(), B, Tq, H, dh, Lk, 1 if causal else 0, pos0, L.stream())
    L.check(rc, 'm3p_attn_query_fwd')
    return ctx


def attn_rows_fwd(q, kv, klen, B, Tq, H, dh, Lk, causal=False, seed=0, p_drop=0.0):
    """Training form of attn_query_fwd: dropout on the probabilities, returns (ctx bf16 [B*Tq, H*dh], lse fp32 [B, H, Tq])."""
    _chk_bf16(q, kv)
    assert q.stride(1) == 1 and kv.dim() == 3 and kv.stride(2) == 1 and kv.shape[0] == B and kv.shape[1] >= Lk
    ctx = torch.empty((B * Tq, H * dh), dtype=BF16, device=q.device)
    lse = torch.empty((B, H, Tq), dtype=torch.float32, device=q.device)
    rc = L.load().m3p_attn_rows_fwd(q.data_ptr(), q.stride(0), kv.data_ptr(), kv.stride(0), kv.stride(1), L.ptr(klen),
                                    ctx.data_ptr(), lse.data_ptr(), B, Tq, H, dh, Lk, 1 if causal else 0, 0, seed,
                                    L.thresh24(p_drop), 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0, L.stream())
    L.check(rc, 'm3p_attn_rows_fwd')
    return ctx, lse


def attn_rows_bwd(q, kv, klen, dctx, lse, B, Tq, H, dh, Lk, qscale, causal=False, seed=0, p_drop=0.0, dq_out=None):
    """-> (dq bf16 [B*Tq, H*dh] (or written into dq_out, row pitch dq_out.stride(0)), dkv fp32 [B, Lk, 2*H*dh])."""
    _chk_bf16(q, kv, dctx)
    assert dctx.is_contiguous()
    d = H * dh
    dq = dq_out if dq_out is not None else torch.empty((B * Tq, d), dtype=BF16, device=q.device)
    dkv = torch.zeros((B, Lk, 2 * d), dtype=torch.float32, device=q.device)
    rc = L.load().m3p_attn_rows_bwd(q.data_ptr(), q.stride(0), kv.data_ptr(), kv.stride(0), kv.stride(1), L.ptr(klen),
                                    dctx.data_ptr(), lse.data_ptr(), dq.data_ptr(), dq.stride(0), dkv.data_ptr(), B, Tq, H, dh, Lk,
                                    1 if causal else 0, 0, qscale, seed, L.thresh24(p_drop),
                                    1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0, L.stream())
    L.check(rc, 'm3p_attn_rows_bwd')
    return dq, dkv


def cast_bf16(x):
    """fp32 -> bf16 through the HIP cast kernel (bf16 input is returned unchanged)."""
    if x.dtype == BF16:
        return x
    assert x.dtype == torch.float32 and x.is_contiguous() and x.numel() % 4 == 0
    out = torch.empty(x.shape, dtype=BF16, device=x.device)
    L.check(L.load().m3p_cast_f32_bf16(x.data_ptr(), out.data_ptr(), x.numel(), L.stream()), 'm3p_cast_f32_bf16')
    return out


_TILE_QUEUE = {}      # device index -> the counter ring handed to m3p_set_tile_queue (kept alive here)


def set_tile_queue(enable=True, slots=4096):
    """Dynamic tile queues for the persistent NT GEMM (include/m3p_hip.h: m3p_set_tile_queue): workgroups pop output tiles
    from per-XCD queues, so CUs slowed down by a co-resident collective kernel take fewer tiles.  Process-wide; all GEMMs on
    one stream."""
    dev = torch.cuda.current_device()
    if not enable:
        L.check(L.load().m3p_set_tile_queue(None, 0), 'm3p_set_tile_queue')
        _TILE_QUEUE.pop(dev, None)
        return
    pool = torch.zeros(slots * 8, dtype=torch.int32, device='cuda')
    torch.cuda.current_stream().synchronize()
    L.check(L.load().m3p_set_tile_queue(pool.data_ptr(), slots), 'm3p_set_tile_queue')
    _TILE_QUEUE[dev] = pool


def cast_f32_bf16_into(src, dst):
    """dst (bf16) <- src (fp32), same element count, on the current stream."""
    assert src.dtype == torch.float32 and dst.dtype == BF16 and src.is_contiguous() and dst.is_contiguous()
    assert src.numel() == dst.numel() and src.numel() % 4 == 0
    L.check(L.load().m3p_cast_f32_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), L.stream()), 'm3p_cast_f32_bf16')


def cast_rows_bf16(x):
    """fp32 (n0, n1, cols) with ANY leading strides (unit stride along cols) -> contiguous bf16 [n0 * n1, cols]: the cast
    reads through a transposed view (the collate's (n, R, 2048) features seen as (R, n, 2048)) instead of copying it first."""
    assert x.dim() == 3 and x.dtype == torch.float32 and x.stride(2) == 1
    n0, n1, cols = x.shape
    out = torch.empty((n0 * n1, cols), dtype=BF16, device=x.device)
    L.check(L.load().m3p_cast_rows_f32_bf16(x.data_ptr(), x.stride(0), x.stride(1), n0, n1, cols, out.data_ptr(), L.stream()),
            'm3p_cast_rows_f32_bf16')
    return out


def seq_masks(lengths, lengths_b, B, S):
    """-> (totlen int32 [B], rowmask uint8 [B * S]): totlen = lengths (+ lengths_b), rowmask[b, s] = s < totlen[b]."""
    dev = lengths.device
    assert lengths.dtype == torch.int64 and lengths.is_contiguous() and lengths.numel() == B
    if lengths_b is not None:
        assert lengths_b.dtype == torch.int64 and lengths_b.is_contiguous() and lengths_b.numel() == B and lengths_b.device == dev
    totlen = torch.empty((B,), dtype=torch.int32, device=dev)
    rowmask = torch.empty((B * S,), dtype=torch.uint8, device=dev)
    L.check(L.load().m3p_seq_masks(lengths.data_ptr(), L.ptr(lengths_b), B, S, totlen.data_ptr(), rowmask.data_ptr(), L.stream()),
            'm3p_seq_masks')
    return totlen, rowmask


def mask_to_rows(mask, inner, s0, s1, soff, d, n_rows):
    """Row numbers (int32 [n_rows]) of the True entries of ``mask`` (bool / uint8, flat order t * inner + b) inside the
    [*, d] row buffer under a strided (T, inner, d) view: (soff + t * s0 + b * s1) // d."""
    m = mask.reshape(-1)
    if m.dtype == torch.bool:
        m = m.view(torch.uint8)
    assert m.dtype == torch.uint8 and m.is_contiguous()
    rows = torch.empty((n_rows,), dtype=torch.int32, device=m.device)
    if _DEBUG_CHECKS:      # (M3P_DEBUG_CHECKS=1: a host sync per call - fewer True entries than n_rows would point the tail at row 0)
        assert int(m.sum().item()) == n_rows, 'mask holds %d True entries, the caller counted %d' % (int(m.sum().item()), n_rows)
    L.check(L.load().m3p_mask_to_rows(m.data_ptr(), m.numel(), inner, s0, s1, soff, d, rows.data_ptr(), n_rows, L.stream()),
            'm3p_mask_to_rows')
    return rows


def scale_bf16_dev(x, g):
    """bf16(g[0] * x) for a bf16 or fp32 ``x`` and a device scalar ``g`` (fp32 [1])."""
    assert x.is_contiguous() and x.dtype in (BF16, torch.float32) and g.dtype == torch.float32 and g.numel() == 1
    out = torch.empty(x.shape, dtype=BF16, device=x.device)
    L.check(L.load().m3p_scale_bf16_dev(x.data_ptr(), int(x.dtype == torch.float32), g.data_ptr(), out.data_ptr(), x.numel(),
                                        L.stream()), 'm3p_scale_bf16_dev')
    return out


def axpy_dev(dst, src, g):
    """dst += g[0] * src (fp32, contiguous), g a device scalar."""
    assert dst.dtype == src.dtype == g.dtype == torch.float32 and dst.is_contiguous() and src.is_contiguous()
    assert dst.numel() == src.numel()
    L.check(L.load().m3p_axpy_dev_f32(dst.data_ptr(), src.data_ptr(), g.data_ptr(), dst.numel(), L.stream()), 'm3p_axpy_dev_f32')


def itm_loss_fwd_bwd(scores, pos, sample_n, w_ce, w_bce):
    """-> (loss fp32 [1], dscores fp32 like scores): xtrainer.py:2357-2372 (CE over groups of sample_n + BCE vs one-hot)."""
    sc = scores.reshape(-1)
    assert sc.dtype == torch.float32 and sc.is_contiguous() and pos.dtype == torch.int64 and pos.is_contiguous()
    G = pos.numel()
    assert sc.numel() == G * sample_n, (sc.numel(), G, sample_n)
    loss = torch.empty((1,), dtype=torch.float32, device=sc.device)
    dsc = torch.empty_like(sc)
    L.check(L.load().m3p_itm_loss_fwd_bwd(sc.data_ptr(), pos.data_ptr(), G, sample_n, float(w_ce), float(w_bce), loss.data_ptr(),
                                          dsc.data_ptr(), L.stream()), 'm3p_itm_loss_fwd_bwd')
    return loss, dsc


def _drop_args(p):
    return L.thresh24(p), (1.0 / (1.0 - p) if p > 0 else 1.0)


def embed_assemble_fwd(tok, emb16, pos, img_proj, loc, w_loc, b_loc, g_img, be_img, g_emb, be_emb, totlen,
                       B, T, R, d, seed_img=0, seed_emb=0, p_drop=0.0, img_rows=None, img_saved=None):
    """img_rows (bf16 [B*R, d], rows b*R + r) replaces the computed image rows (AoA refiner output); img_saved is
    then the (e, mean_i, rstd_i) triple embed_image_rows_fwd returned for backward."""
    dev = emb16.device
    S = R + T
    h = torch.empty((B * S, d), dtype=BF16, device=dev)
    z = torch.empty((B * S, d), dtype=BF16, device=dev)
    mean_e = torch.empty(B * S, dtype=torch.float32, device=dev)
    rstd_e = torch.empty(B * S, dtype=torch.float32, device=dev)
    if img_rows is not None:
        assert img_rows.dtype == BF16 and img_rows.is_contiguous() and img_rows.shape == (B * R, d)
        e, mean_i, rstd_i = img_saved
    else:
        e = torch.empty((max(R * B, 1), d), dtype=BF16, device=dev)
        mean_i = torch.empty(max(R * B, 1), dtype=torch.float32, device=dev)
        rstd_i = torch.empty(max(R * B, 1), dtype=torch.float32, device=dev)
    th, ik = _drop_args(p_drop)
    rc = L.load().m3p_embed_assemble_fwd(
        tok.data_ptr(), emb16.data_ptr(), pos.data_ptr(), L.ptr(img_proj), L.ptr(loc), w_loc.data_ptr(),
        b_loc.data_ptr(), g_img.data_ptr(), be_img.data_ptr(), g_emb.data_ptr(), be_emb.data_ptr(), totlen.data_ptr(),
        h.data_ptr(), z.data_ptr(), mean_e.data_ptr(), rstd_e.data_ptr(), e.data_ptr(), mean_i.data_ptr(),
        rstd_i.data_ptr(), B, T, R, d, seed_img, seed_emb, th, ik, L.ptr(img_rows), L.stream())
    L.check(rc, 'm3p_embed_assemble_fwd')
    return h, (z, mean_e, rstd_e, e, mean_i, rstd_i)


def embed_image_rows_fwd(img_proj, loc, w_loc, b_loc, g_img, be_img, B, R, d, seed_img=0, p_drop=0.0):
    """Image rows after LayerNorm + dropout as their own tensor (bf16 [B*R, d], rows b*R + r) and the saved
    (e, mean_i, rstd_i) for backward."""
    dev = img_proj.device
    e = torch.empty((R * B, d), dtype=BF16, device=dev)
    mean_i = torch.empty(R * B, dtype=torch.float32, device=dev)
    rstd_i = torch.empty(R * B, dtype=torch.float32, device=dev)
    rows = torch.empty((B * R, d), dtype=BF16, device=dev)
    th, ik = _drop_args(p_drop)
    rc = L.load().m3p_embed_image_rows_fwd(img_proj.data_ptr(), loc.data_ptr(), w_loc.data_ptr(), b_loc.data_ptr(),
                                           g_img.data_ptr(), be_img.data_ptr(), e.data_ptr(), mean_i.data_ptr(),
                                           rstd_i.data_ptr(), rows.data_ptr(), B, R, d, seed_img, th, ik, L.stream())
    L.check(rc, 'm3p_embed_image_rows_fwd')
    return rows, (e, mean_i, rstd_i)


def embed_image_rows_bwd(d_rows, img_saved, g_img, loc, totlen, grads, B, R, d, seed_img=0, p_drop=0.0):
    """Backward of embed_image_rows_fwd alone (the image-only stream): d_rows bf16 [B*R, d] = gradient wrt the rows it
    returned -> its dropout, LayerNorm and location-projection backward (the image half of m3p_embed_assemble_bwd with no
    token rows).  grads: d_g_img, d_be_img, d_b_img, d_b_loc, d_w_loc.  Returns de (bf16 [R*B, d], rows r*B + b): the
    gradient wrt the image projection's output."""
    e, mean_i, rstd_i = img_saved
    assert d_rows.dtype == BF16 and d_rows.is_contiguous() and d_rows.shape == (B * R, d)
    de = torch.empty_like(e)
    th, ik = _drop_args(p_drop)
    dummy = d_rows.data_ptr()       # the token-row arguments are not read by the image half
    args = (dummy, dummy, mean_i.data_ptr(), rstd_i.data_ptr(), g_img.data_ptr(), e.data_ptr(), mean_i.data_ptr(),
            rstd_i.data_ptr(), g_img.data_ptr(), dummy, totlen.data_ptr(), loc.data_ptr(), d_rows.data_ptr(), de.data_ptr(),
            grads['d_g_img'].data_ptr(), grads['d_be_img'].data_ptr(), grads['d_g_img'].data_ptr(), grads['d_g_img'].data_ptr(),
            None, grads['d_g_img'].data_ptr(), grads['d_be_img'].data_ptr(), grads['d_b_img'].data_ptr(),
            grads['d_b_loc'].data_ptr(), grads['d_w_loc'].data_ptr(), B, 0, R, d, -1, seed_img, 0, th, ik)
    L.check(L.load().m3p_embed_assemble_bwd(*args, 2, L.stream()), 'm3p_embed_assemble_bwd')
    return de


def embed_assemble_bwd(dh, saved, g_emb, g_img, tok, totlen, loc, grads, B, T, R, d, pad_index,
                       seed_img=0, seed_emb=0, p_drop=0.0, img_rows_bwd=None, tok_rows=None):
    """grads: dict of fp32 gradient views (d_g_emb, d_be_emb, d_pos, d_emb, d_g_img, d_be_img, d_b_img,
    d_b_loc, d_w_loc).  Returns de (bf16 [R*B, d]).
    img_rows_bwd (refine_image): callable taking the gradient of the refined image rows (bf16 [B*R, d]) and
    returning the gradient of the refiner's input; run between the two halves of the backward.
    tok_rows (data parallelism): bf16 [T*B, d] buffer that receives the token rows' gradients instead of the
    scatter-add into d_emb."""
    z, mean_e, rstd_e, e, mean_i, rstd_i = saved
    dz = torch.empty_like(z)
    de = torch.empty_like(e)
    th, ik = _drop_args(p_drop)
    if tok_rows is not None:
        assert tok_rows.dtype == BF16 and tok_rows.is_contiguous() and tok_rows.shape == (T * B, d)
    args = (dh.data_ptr(), z.data_ptr(), mean_e.data_ptr(), rstd_e.data_ptr(), g_emb.data_ptr(), e.data_ptr(),
            mean_i.data_ptr(), rstd_i.data_ptr(), g_img.data_ptr(), tok.data_ptr(), totlen.data_ptr(), L.ptr(loc),
            dz.data_ptr(), de.data_ptr(), grads['d_g_emb'].data_ptr(), grads['d_be_emb'].data_ptr(),
            grads['d_pos'].data_ptr(), grads['d_emb'].data_ptr(), L.ptr(tok_rows), grads['d_g_img'].data_ptr(),
            grads['d_be_img'].data_ptr(), grads['d_b_img'].data_ptr(), grads['d_b_loc'].data_ptr(),
            grads['d_w_loc'].data_ptr(), B, T, R, d, pad_index, seed_img, seed_emb, th, ik)
    if img_rows_bwd is not None:
        L.check(L.load().m3p_embed_assemble_bwd(*args, 1, L.stream()), 'm3p_embed_assemble_bwd')
        dz_img = dz.view(B, R + T, d)[:, :R, :]
        dz_img.copy_(img_rows_bwd(dz_img.contiguous().view(B * R, d)).view(B, R, d))
        L.check(L.load().m3p_embed_assemble_bwd(*args, 2, L.stream()), 'm3p_embed_assemble_bwd')
        return de
    L.check(L.load().m3p_embed_assemble_bwd(*args, 0, L.stream()), 'm3p_embed_assemble_bwd')
    return de


def scatter_add_token_rows(rows, ids, dst, pad_index):
    """dst[ids[i], :] (fp32 [V, d]) += rows[i, :] (bf16 [n, d], any row pitch); pad rows skipped."""
    _chk_bf16(rows)
    assert ids.dtype == torch.int64 and dst.dtype == torch.float32 and rows.stride(1) == 1 and ids.is_contiguous()
    n, d = rows.shape
    assert ids.numel() == n and dst.shape[1] == d and dst.stride(0) == d
    L.check(L.load().m3p_scatter_add_token_rows(rows.data_ptr(), rows.stride(0), ids.data_ptr(), dst.data_ptr(), n, d, int(pad_index),
                                                L.stream()), 'm3p_scatter_add_token_rows')


def gather_rows(src_base, idx, n, d):
    out = torch.empty((n, d), dtype=BF16, device=idx.device)
    L.check(L.load().m3p_gather_rows(src_base.data_ptr(), idx.data_ptr(), out.data_ptr(), n, d, L.stream()), 'm3p_gather_rows')
    return out


def scatter_add_rows(src, idx, dst_base, n, d):
    L.check(L.load().m3p_scatter_add_rows(src.data_ptr(), idx.data_ptr(), dst_base.data_ptr(), n, d, L.stream()),
            'm3p_scatter_add_rows')


def ce_fwd_bwd(logits, V, target, loss_scale, grad_scale):
    """In place: logits <- dlogits.  Returns (loss_sum [1] fp32, row_loss [n] fp32)."""
    n, ld = logits.shape[0], logits.stride(0)
    row_loss = torch.empty(n, dtype=torch.float32, device=logits.device)
    assert target.dtype == torch.int64
    rc = L.load().m3p_ce_fwd_bwd(logits.data_ptr(), ld, n, V, target.data_ptr(), row_loss.data_ptr(),
                                 None, loss_scale, grad_scale, L.stream())
    L.check(rc, 'm3p_ce_fwd_bwd')
    loss_sum = (row_loss.sum() * loss_scale).reshape(1)   # tiny reduction; avoids n same-address atomics
    return loss_sum, row_loss


_CE_WS = {}


def ce_fwd_bwd_colsum(logits, V, target, loss_scale, grad_scale):
    """ce_fwd_bwd + the column sums of the gradient from the same pass.  In place: logits <- dlogits.
    Returns (loss_sum [1], row_loss [n], colsum fp32 [ld] of the rounded gradient)."""
    n, ld = logits.shape[0], logits.stride(0)
    dev = logits.device
    row_loss = torch.empty(n, dtype=torch.float32, device=dev)
    row_lse = torch.empty(n, dtype=torch.float32, device=dev)
    cs = torch.empty(ld, dtype=torch.float32, device=dev)
    need = L.load().m3p_ce_colsum_workspace_bytes(ld, n)
    ws = _CE_WS.get(dev)
    if ws is None or ws.numel() < need:
        ws = _CE_WS[dev] = torch.empty(need, dtype=torch.uint8, device=dev)
    assert target.dtype == torch.int64
    rc = L.load().m3p_ce_fwd_bwd_colsum(logits.data_ptr(), ld, n, V, target.data_ptr(), row_loss.data_ptr(), row_lse.data_ptr(),
                                        grad_scale, cs.data_ptr(), ws.data_ptr(), ws.numel(), L.stream())
    L.check(rc, 'm3p_ce_fwd_bwd_colsum')
    return (row_loss.sum() * loss_scale).reshape(1), row_loss, cs


def ce_from_block_stats(logits, V, target, stats, loss_scale, grad_scale):
    """ce_fwd_bwd_colsum with the first pass replaced by the 64-column block statistics the vocabulary projection wrote
    (gemm_nt(..., EPI_BIAS_LSE, out2=stats)): the rows' log-sum-exp is a reduction over N / 64 pairs, not over the logits."""
    n, ld = logits.shape[0], logits.stride(0)
    dev = logits.device
    row_loss = torch.empty(n, dtype=torch.float32, device=dev)
    row_lse = torch.empty(n, dtype=torch.float32, device=dev)
    cs = torch.empty(ld, dtype=torch.float32, device=dev)
    scratch = torch.empty((32, n, 2), dtype=torch.float32, device=dev)
    assert target.dtype == torch.int64 and stats.dtype == torch.float32 and stats.shape[1] == n
    L.check(L.load().m3p_ce_lse_from_blocks(stats.data_ptr(), stats.shape[0], n, logits.data_ptr(), ld, target.data_ptr(),
                                            row_loss.data_ptr(), row_lse.data_ptr(), scratch.data_ptr(), L.stream()),
            'm3p_ce_lse_from_blocks')
    need = L.load().m3p_ce_colsum_workspace_bytes(ld, n)
    ws = _CE_WS.get(dev)
    if ws is None or ws.numel() < need:
        ws = _CE_WS[dev] = torch.empty(need, dtype=torch.uint8, device=dev)
    L.check(L.load().m3p_ce_bwd_colsum(logits.data_ptr(), ld, n, V, target.data_ptr(), row_lse.data_ptr(), grad_scale, cs.data_ptr(),
                                       ws.data_ptr(), ws.numel(), L.stream()), 'm3p_ce_bwd_colsum')
    return (row_loss.sum() * loss_scale).reshape(1), row_loss, cs


def colsum(x, ncols, out, scale=None):
    rc = L.load().m3p_colsum_bf16(x.data_ptr(), x.stride(0), x.shape[0], ncols, out.data_ptr(), L.ptr(scale), L.stream())
    L.check(rc, 'm3p_colsum_bf16')


def sumsq(g, out):
    L.check(L.load().m3p_sumsq_f32(g.data_ptr(), g.numel(), out.data_ptr(), L.stream()), 'm3p_sumsq_f32')


def sumsq_ranges(buf, ranges, out):
    """out += sum of squares over the pieces buf[a:b] for (a, b) in ranges - one launch (m3p_sumsq_ranges_f32)."""
    ranges = [(int(a), int(b)) for a, b in ranges if b > a]
    if not ranges:
        return
    if len(ranges) == 1:
        return sumsq(buf[ranges[0][0]:ranges[0][1]], out)
    n = len(ranges)
    starts = (C.c_longlong * n)(*[a for a, _ in ranges])
    counts = (C.c_longlong * n)(*[b - a for a, b in ranges])
    L.check(L.load().m3p_sumsq_ranges_f32(buf.data_ptr(), starts, counts, n, out.data_ptr(), L.stream()), 'm3p_sumsq_ranges_f32')


def adam_step(p, g, m, v, w16, lr, beta1, beta2, eps, weight_decay, step_size, gnorm_sq=None, max_norm=0.0,
              grad_scale=1.0, zero_grad=True):
    n = p.numel()
    rc = L.load().m3p_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), L.ptr(w16), n, lr, beta1, beta2,
                                eps, weight_decay, step_size, L.ptr(gnorm_sq), max_norm, grad_scale, int(zero_grad),
                                L.stream())
    L.check(rc, 'm3p_adam_step')


def adam_step_ranges(p, g, m, v, w16, pieces, lr, beta1, beta2, eps, weight_decay, gnorm_sq=None, max_norm=0.0, grad_scale=1.0):
    """One launch of the fused Adam update over several pieces of the flat arenas: pieces = [(start, end, step_size, zero_grad)]."""
    pieces = [q for q in pieces if q[1] > q[0]]
    if not pieces:
        return
    n = len(pieces)
    starts = (C.c_longlong * n)(*[int(a) for a, _, _, _ in pieces])
    counts = (C.c_longlong * n)(*[int(b - a) for a, b, _, _ in pieces])
    steps = (C.c_float * n)(*[float(st) for _, _, st, _ in pieces])
    zeros = (C.c_int * n)(*[int(bool(z)) for _, _, _, z in pieces])
    rc = L.load().m3p_adam_step_ranges(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), L.ptr(w16), starts, counts, steps, zeros, n,
                                       lr, beta1, beta2, eps, weight_decay, L.ptr(gnorm_sq), max_norm, grad_scale, L.stream())
    L.check(rc, 'm3p_adam_step_ranges')


def transpose_bf16(src, dst):
    """dst[c, r] = src[r, c]; dst may have a row pitch larger than rows (pad columns untouched)."""
    rows, cols = src.shape
    rc = L.load().m3p_transpose_bf16(src.data_ptr(), dst.data_ptr(), rows, cols, src.stride(0), dst.stride(0), L.stream())
    L.check(rc, 'm3p_transpose_bf16')


def gelu_fwd(u, grad_inplace=False):
    """h = gelu_erf(u).  With grad_inplace, u is overwritten by gelu_erf'(u) (bf16) for EPI_MUL in backward."""
    h = torch.empty_like(u)
    L.check(L.load().m3p_gelu_fwd(u.data_ptr(), h.data_ptr(), u.data_ptr() if grad_inplace else None, u.numel(), L.stream()),
            'm3p_gelu_fwd')
    return h


GQ_OFF, GQ_STEP = 27.0 / 201.0, 1.0 / 201.0       # the byte code of gelu' (csrc/common.hpp): gelu' ~ code * GQ_STEP - GQ_OFF


def gq_eligible(M, N):
    """Shapes whose FFN runs on the byte-derivative form: whole 256 x 256 tiles of the eight-wave kernel."""
    return M >= 1024 and M % 256 == 0 and N % 256 == 0 and N >= 512


def gelu_fwd_gq(u):
    """(h = gelu_erf(u) bf16 [M, N], gq uint8 [M * N]): gq holds gelu_erf'(u) as one byte per element in the fragment
    order of the eight-wave GEMM (include/m3p_hip.h: m3p_gelu_fwd_gq) - what gemm_nt(..., EPI_MULQ, aux=gq) multiplies by.
    u is dead afterwards."""
    _chk_bf16(u)
    M, N = u.shape
    assert u.is_contiguous() and gq_eligible(M, N), (M, N)
    h = torch.empty_like(u)
    gq = torch.empty((M * N,), dtype=torch.uint8, device=u.device)
    L.check(L.load().m3p_gelu_fwd_gq(u.data_ptr(), h.data_ptr(), gq.data_ptr(), M, N, L.stream()), 'm3p_gelu_fwd_gq')
    return h, gq


def gq_unpack(gq, M, N):
    """The decoded derivative as a float [M, N] matrix in row order (tests): inverse of the fragment-order layout."""
    t = gq.view(M // 256, N // 256, 2, 4, 8, 4, 16, 4, 4)       # tm, tn, wm, wn, i, fg, fr, j, r
    t = t.permute(0, 2, 4, 6, 1, 3, 7, 5, 8).reshape(M, N)     # rows: tm, wm, i, fr   cols: tn, wn, j, fg, r
    return t.float() * GQ_STEP - GQ_OFF


def gelu_fwd_q8(u, scale, amax=None):
    """(h = gelu_erf(u) bf16, h8 = e4m3(scale * h) uint8) in one pass; amax (fp32 [1]) is raised to max |h|."""
    h = torch.empty_like(u)
    h8 = torch.empty(u.shape, dtype=torch.uint8, device=u.device)
    L.check(L.load().m3p_gelu_fwd_q8(u.data_ptr(), h.data_ptr(), h8.data_ptr(), u.numel(), scale.data_ptr(), L.ptr(amax), L.stream()),
            'm3p_gelu_fwd_q8')
    return h, h8


def transpose_batch(desc, n_desc, max_tiles):
    L.check(L.load().m3p_transpose_batch_bf16(desc.data_ptr(), n_desc, max_tiles, L.stream()), 'm3p_transpose_batch_bf16')


def itm_head_fwd(first, W1_16, b1, w2, b2):
    """first: bf16 [B, d] (hidden[:, 0]); W1_16: bf16 [d, d] working copy of pooled_layer.dense.weight.
    Returns (h16 contiguous bf16 copy of the input rows, pooled fp32 [B, d], scores fp32 [B])."""
    _chk_bf16(first, W1_16)
    B, d = first.shape
    h16 = first.contiguous()
    pre = gemm_nt(h16, W1_16, L.EPI_BIAS, bias=b1)
    pooled = torch.empty((B, d), dtype=torch.float32, device=first.device)
    scores = torch.empty((B,), dtype=torch.float32, device=first.device)
    rc = L.load().m3p_itm_score_fwd(pre.data_ptr(), w2.data_ptr(), b2.data_ptr(), pooled.data_ptr(), scores.data_ptr(),
                                    B, d, L.stream())
    L.check(rc, 'm3p_itm_score_fwd')
    return h16, pooled, scores


def itm_head_bwd(dscores, h16, pooled, W1_16, w2, dW1, db1, dw2, db2):
    """Returns dh bf16 [B, d]; accumulates dW1 [d,d], db1, dw2 [d], db2 [1] (fp32) in place."""
    B, d = h16.shape
    dev = h16.device
    ldt = (B + 7) // 8 * 8
    dpre16 = torch.empty((B, d), dtype=BF16, device=dev)
    dpreT16 = torch.zeros((d, ldt), dtype=BF16, device=dev)
    assert dscores.dtype == torch.float32 and dscores.is_contiguous() and dscores.numel() == B
    rc = L.load().m3p_itm_score_bwd(dscores.data_ptr(), pooled.data_ptr(), w2.data_ptr(), dpre16.data_ptr(),
                                    dpreT16.data_ptr(), ldt, db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(), B, d, L.stream())
    L.check(rc, 'm3p_itm_score_bwd')
    gemm_wgrad(dpre16, h16, dW1)                         # dW1[j][k] += sum_b dpre[b][j] h[b][k]
    dh32 = torch.zeros((B, d), dtype=torch.float32, device=dev)
    gemm_wgrad(dpreT16, W1_16, dh32, n=B)                # dh[b][k]  = sum_j dpre[b][j] W1[j][k]
    return dh32.to(BF16)


def gelu_bwd(dy, u):
    """du = dy * gelu_erf'(u) (bf16)."""
    _chk_bf16(dy, u)
    assert dy.shape == u.shape and dy.is_contiguous() and u.is_contiguous()
    du = torch.empty_like(dy)
    L.check(L.load().m3p_gelu_bwd(dy.data_ptr(), u.data_ptr(), du.data_ptr(), dy.numel(), L.stream()), 'm3p_gelu_bwd')
    return du


def mse_fwd_bwd(pred, tgt, grad_scale):
    """pred bf16 [n, c], tgt fp32 [n, c] -> (sum of squared errors as a 1-element fp32 tensor, dpred bf16 = 2 (pred - tgt) grad_scale)."""
    _chk_bf16(pred)
    n, c = pred.shape
    assert tgt.shape == (n, c) and tgt.dtype == torch.float32 and pred.stride(1) == 1 and tgt.stride(1) == 1
    dpred = torch.empty_like(pred)
    row_sq = torch.empty((n,), dtype=torch.float32, device=pred.device)
    rc = L.load().m3p_mse_fwd_bwd(pred.data_ptr(), pred.stride(0), tgt.data_ptr(), tgt.stride(0), dpred.data_ptr(),
                                  row_sq.data_ptr(), n, c, grad_scale, L.stream())
    L.check(rc, 'm3p_mse_fwd_bwd')
    return row_sq.sum().reshape(1), dpred



def dropout_rows(x, p_drop, seed, res=None, out=None, rng_ld=None, rng_col0=0):
    """out = (res or 0) + dropout(x) on a 2-D bf16 view (last dim contiguous; x / res / out may be column slices of
    wider buffers).  The keep bit of element (r, c) is rng.keep(r * rng_ld + rng_col0 + c): rng_ld defaults to
    x.shape[1].  out may be x."""
    _chk_bf16(x)
    rows, cols = x.shape
    assert x.stride(1) == 1 and (res is None or (res.dtype == BF16 and res.stride(1) == 1 and res.shape == x.shape))
    if out is None:
        out = torch.empty((rows, cols), dtype=BF16, device=x.device)
    assert out.dtype == BF16 and out.stride(1) == 1 and out.shape == x.shape
    th, ik = _drop_args(p_drop)
    rc = L.load().m3p_dropout_rows(x.data_ptr(), x.stride(0), L.ptr(res), res.stride(0) if res is not None else 0,
                                   out.data_ptr(), out.stride(0), rows, cols, cols if rng_ld is None else rng_ld, rng_col0,
                                   seed, th, ik, L.stream())
    L.check(rc, 'm3p_dropout_rows')
    return out


def glu_fwd(ab):
    """nn.GLU: ab bf16 [rows, 2d] -> ab[:, :d] * sigmoid(ab[:, d:])."""
    _chk_bf16(ab)
    rows, d2 = ab.shape
    assert ab.is_contiguous() and d2 % 2 == 0
    y = torch.empty((rows, d2 // 2), dtype=BF16, device=ab.device)
    L.check(L.load().m3p_glu_fwd(ab.data_ptr(), d2, y.data_ptr(), rows, d2 // 2, L.stream()), 'm3p_glu_fwd')
    return y


def glu_bwd(ab, dy):
    _chk_bf16(ab, dy)
    rows, d2 = ab.shape
    assert ab.is_contiguous() and dy.is_contiguous() and dy.shape == (rows, d2 // 2)
    dab = torch.empty_like(ab)
    L.check(L.load().m3p_glu_bwd(ab.data_ptr(), d2, dy.data_ptr(), dab.data_ptr(), rows, d2 // 2, L.stream()), 'm3p_glu_bwd')
    return dab
