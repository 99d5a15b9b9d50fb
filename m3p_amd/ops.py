"""Raw (non-autograd) launchers: one Python function per C-ABI entry point.

Each allocates its outputs with torch (device memory + caching allocator are plumbing),
passes raw device pointers and the current HIP stream across the C ABI and raises on any
error code.  The autograd layer (m3p_amd/functional.py) composes these."""
import ctypes as C

import torch

from . import lib as L

BF16 = torch.bfloat16


def _chk_bf16(*ts):
    for t in ts:
        if t is not None:
            assert t.dtype == BF16 and t.is_cuda, (t.dtype, t.device)


def gemm_nt(a, w, epilogue=L.EPI_NONE, bias=None, aux=None, out=None, out2=None, colsum=None,
            scale_cols=0, scale=1.0, alpha=1.0, seed=0, p_drop=0.0, n=None):
    """C[M,N] = epi(a[M,K] @ w[N,K]^T).  a, w bf16 (row pitch = stride(0)); returns C (bf16).
    ``n`` restricts the number of output columns (rows of w) used."""
    _chk_bf16(a, w, aux, out, out2)
    M, K = a.shape
    N = w.shape[0] if n is None else n
    assert w.shape[1] == K and a.stride(1) == 1 and w.stride(1) == 1
    if out is None:
        out = torch.empty((M, N), dtype=BF16, device=a.device)
    ep = L.Epilogue()
    ep.bias = L.ptr(bias)
    ep.aux = L.ptr(aux)
    ep.out2 = L.ptr(out2)
    ep.colsum = L.ptr(colsum)
    ep.ld_aux = aux.stride(0) if aux is not None else 0
    ep.ld_out2 = out2.stride(0) if out2 is not None else 0
    ep.scale_cols = scale_cols
    ep.scale = scale
    ep.alpha = alpha
    ep.seed = seed
    ep.thresh24 = L.thresh24(p_drop)
    ep.inv_keep = 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0
    if bias is not None:
        assert bias.dtype == torch.float32
    rc = L.load().m3p_gemm_nt_bf16(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(),
                                   out.stride(0), M, N, K, epilogue, C.byref(ep), L.stream())
    L.check(rc, 'm3p_gemm_nt_bf16')
    return out


def gemm_wgrad(dy, x, dw, alpha=1.0, n=None, k=None):
    """dw[N,K] (fp32) += alpha * dy[M,N]^T @ x[M,K]."""
    _chk_bf16(dy, x)
    assert dw.dtype == torch.float32 and dw.stride(-1) == 1
    M = dy.shape[0]
    N = dy.shape[1] if n is None else n
    K = x.shape[1] if k is None else k
    assert x.shape[0] == M and dw.shape[0] >= N and dw.shape[1] >= K
    rc = L.load().m3p_gemm_wgrad_bf16(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dw.data_ptr(),
                                      dw.stride(0), M, N, K, alpha, L.stream())
    L.check(rc, 'm3p_gemm_wgrad_bf16')
    return dw


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


def attn_fwd(qkv, keylen, B, S, H, dh, seed=0, p_drop=0.0):
    """qkv bf16 [B*S, 3*H*dh] -> (ctx bf16 [B*S, H*dh], lse fp32 [B,H,S])."""
    _chk_bf16(qkv)
    assert qkv.is_contiguous() and qkv.shape == (B * S, 3 * H * dh) and keylen.dtype == torch.int32
    ctx = torch.empty((B * S, H * dh), dtype=BF16, device=qkv.device)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=qkv.device)
    rc = L.load().m3p_attn_fwd(qkv.data_ptr(), keylen.data_ptr(), ctx.data_ptr(), lse.data_ptr(), B, S, H, dh,
                               seed, L.thresh24(p_drop), 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0, L.stream())
    L.check(rc, 'm3p_attn_fwd')
    return ctx, lse


def attn_bwd(qkv, keylen, ctx, dctx, lse, B, S, H, dh, dbias_qkv=None, seed=0, p_drop=0.0):
    _chk_bf16(qkv, ctx, dctx)
    assert dctx.is_contiguous() and ctx.is_contiguous()
    dqkv = torch.empty_like(qkv)
    rc = L.load().m3p_attn_bwd(qkv.data_ptr(), keylen.data_ptr(), ctx.data_ptr(), dctx.data_ptr(), lse.data_ptr(),
                               dqkv.data_ptr(), L.ptr(dbias_qkv), B, S, H, dh, 1.0 / (dh ** 0.5), seed,
                               L.thresh24(p_drop), 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0, L.stream())
    L.check(rc, 'm3p_attn_bwd')
    return dqkv
