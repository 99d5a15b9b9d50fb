import math

import numpy as np
import pytest
import torch

from tests.util import randn_bf16, randn_f32, rel_l2, max_abs

pytestmark = pytest.mark.gpu


def _ref_attention(qkv, keylen, B, S, H, dh, keep=None, p=0.0):
    """fp32 restatement of transformer.py:197-205 on an already-projected, q-prescaled qkv."""
    d = H * dh
    q, k, v = qkv.view(B, S, 3, H, dh).permute(2, 0, 3, 1, 4)      # (B,H,S,dh)
    scores = q @ k.transpose(2, 3)
    mask = torch.arange(S)[None, :] < keylen[:, None]
    scores = scores.masked_fill(~mask[:, None, None, :], float('-inf'))
    w = torch.softmax(scores, dim=-1)
    lse = torch.logsumexp(scores, dim=-1)
    if keep is not None:
        w = w * keep / (1 - p)
    ctx = (w @ v).transpose(1, 2).reshape(B * S, d)
    return ctx, lse


CASES = [(2, 74, 4, 32), (3, 164, 12, 64), (2, 16, 2, 64), (1, 116, 12, 64), (2, 200, 2, 64), (1, 356, 16, 64), (2, 33, 1, 32),
         (1, 512, 2, 64), (2, 1, 2, 64), (1, 500, 3, 32)]     # the maximum sequence, a single position, long with dh = 32


@pytest.mark.parametrize('B,S,H,dh', CASES)
@pytest.mark.parametrize('p', [0.0, 0.1])
def test_attention_fwd_bwd(B, S, H, dh, p):
    from m3p_amd import ops, rng
    d = H * dh
    seed = 4242
    qkv, qkvc = randn_bf16((B * S, 3 * d), 1, 0.7)
    rs = np.random.RandomState(3)
    keylen = torch.from_numpy(rs.randint(max(S // 2, 1), S + 1, size=B).astype(np.int32))
    keylen[0] = S
    ctx, lse = ops.attn_fwd(qkv, keylen.cuda(), B, S, H, dh, seed=seed, p_drop=p)
    keep = None
    if p > 0:
        keep = torch.from_numpy(rng.keep_mask(B * H * S * S, seed, p, (B, H, S, S))).float()
    x = qkvc.clone().requires_grad_(True)
    ctx_ref, lse_ref = _ref_attention(x, keylen.long(), B, S, H, dh, keep, p)
    assert rel_l2(ctx.float(), ctx_ref) < 6e-3
    assert max_abs(lse, lse_ref) < 2e-3
    dctx, dctxc = randn_bf16((B * S, d), 7)
    dbias = torch.zeros(3 * d, device='cuda')
    kmask = None
    if p > 0:
        # the keep words the forward pass leaves for backward (required with dropout on since round 6): same outputs as without
        ctx2, lse2, kmask = ops.attn_fwd(qkv, keylen.cuda(), B, S, H, dh, seed=seed, p_drop=p, want_mask=True)
        assert torch.equal(ctx2, ctx) and torch.equal(lse2, lse)
        with pytest.raises(AssertionError):
            ops.attn_bwd(qkv, keylen.cuda(), ctx, dctx, lse, B, S, H, dh, dbias_qkv=dbias, seed=seed, p_drop=p)
    dqkv = ops.attn_bwd(qkv, keylen.cuda(), ctx, dctx, lse, B, S, H, dh, dbias_qkv=dbias, seed=seed, p_drop=p, keepmask=kmask)
    ctx_ref.backward(dctxc)
    g = x.grad.clone()
    g[:, :d] *= 1.0 / math.sqrt(dh)       # kernel returns the gradient of the unscaled q projection
    gall = float(g.norm())
    for name, sl in (('dq', slice(0, d)), ('dk', slice(d, 2 * d)), ('dv', slice(2 * d, 3 * d))):
        if float(g[:, sl].norm()) < 1e-6 * gall:      # a single key: softmax is constant, dq = dk = 0 exactly
            # (ours: D = rowsum(dO * O) uses the bf16-rounded O, so with dropout's 1/(1-p) scale a 2^-9 residue remains)
            assert float(dqkv[:, sl].float().norm()) < 1e-2 * gall, name
            continue
        assert rel_l2(dqkv[:, sl].float(), g[:, sl]) < 1.5e-2, name
    cs = dqkv.float().sum(0)
    assert rel_l2(dbias[:d], cs[:d]) < 1e-4 and rel_l2(dbias[2 * d:], cs[2 * d:]) < 1e-4   # column sums of the rows it wrote
    # k-bias: softmax shift invariance makes the true gradient 0 (the fp32 reference gives ~1e-7 noise);
    # the kernel writes exact zeros rather than the bf16 rounding noise of its dK rows
    assert bool((dbias[d:2 * d] == 0).all())
    assert float(g[:, d:2 * d].sum(0).abs().max()) <= 1e-3 * float(g[:, d:2 * d].abs().sum(0).max())


def test_attention_perf_smoke():
    from m3p_amd import ops
    B, S, H, dh = 256, 164, 12, 64
    d = H * dh
    qkv, _ = randn_bf16((B * S, 3 * d), 1, 0.7)
    keylen = torch.full((B,), S, dtype=torch.int32, device='cuda')
    for p in (0.0, 0.1):
        ctx, lse, km = ops.attn_fwd(qkv, keylen, B, S, H, dh, seed=1, p_drop=p, want_mask=True)
        dctx = torch.randn_like(ctx); dbias = torch.zeros(3 * d, device="cuda")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.attn_fwd(qkv, keylen, B, S, H, dh, seed=1, p_drop=p)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        fl = 4.0 * B * H * S * S * dh
        print('attn_fwd p=%.1f: %.3f ms  %.1f TF' % (p, ms, fl / ms / 1e9))
        e0.record()
        for _ in range(10):
            ops.attn_bwd(qkv, keylen, ctx, dctx, lse, B, S, H, dh, dbias_qkv=dbias, seed=1, p_drop=p, keepmask=km)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print('attn_bwd p=%.1f: %.3f ms  %.1f TF (algorithmic 2x fwd)' % (p, ms, 2 * fl / ms / 1e9))


@pytest.mark.parametrize('B,S', [(32, 164), (5, 161), (4, 176), (3, 170), (256, 164)])      # 11 tiles / 6 steps: every length from 161 to 176; (256, 164) = the benchmarked launch, twelve heads per persistent workgroup
@pytest.mark.parametrize('p', [0.0, 0.1])
def test_attention_bwd_forms_for_the_m3p_sequence(p, B, S):
    """The three backward forms for 36 regions + 128 tokens (m3p_debug_attn_variant: 1 = two phases with the scores recomputed,
    2 = one pass - dS^T handed from phase A to phase B through LDS - as one twelve-wave workgroup per head, 0 = the same as a
    persistent kernel, the default) on MORE heads than CUs, so that persistent workgroups walk several heads: each against
    the fp32 reference at the kernel's usual bar, and against each other to bf16 rounding."""
    from m3p_amd import ops, rng, lib as L
    H, dh = 12, 64
    d = H * dh
    seed = 99
    qkv, qkvc = randn_bf16((B * S, 3 * d), 11, 0.7)
    rs = np.random.RandomState(5)
    keylen = torch.from_numpy(rs.randint(S // 2, S + 1, size=B).astype(np.int32))
    keylen[0] = S
    kw = dict(seed=seed, p_drop=p)
    if p > 0:
        ctx, lse, kmask = ops.attn_fwd(qkv, keylen.cuda(), B, S, H, dh, want_mask=True, **kw)
        keep = torch.from_numpy(rng.keep_mask(B * H * S * S, seed, p, (B, H, S, S))).float()
    else:
        (ctx, lse), kmask, keep = ops.attn_fwd(qkv, keylen.cuda(), B, S, H, dh, **kw), None, None
    x = qkvc.clone().requires_grad_(True)
    ctx_ref, _ = _ref_attention(x, keylen.long(), B, S, H, dh, keep, p)
    dctx, dctxc = randn_bf16((B * S, d), 13)
    ctx_ref.backward(dctxc)
    g = x.grad.clone()
    g[:, :d] *= 1.0 / math.sqrt(dh)
    outs = {}
    lib = L.load()
    try:
        for variant in (1, 2, 0):
            lib.m3p_debug_attn_variant(variant)
            dbias = torch.zeros(3 * d, device='cuda')
            dqkv = ops.attn_bwd(qkv, keylen.cuda(), ctx, dctx, lse, B, S, H, dh, dbias_qkv=dbias, keepmask=kmask, **kw)
            torch.cuda.synchronize()
            for name, sl in (('dq', slice(0, d)), ('dk', slice(d, 2 * d)), ('dv', slice(2 * d, 3 * d))):
                assert rel_l2(dqkv[:, sl].float(), g[:, sl]) < 1.5e-2, (variant, name)
            cs = dqkv.float().sum(0)
            assert rel_l2(dbias[:d], cs[:d]) < 1e-4 and rel_l2(dbias[2 * d:], cs[2 * d:]) < 1e-4 and bool((dbias[d:2 * d] == 0).all()), variant
            outs[variant] = dqkv.float()
    finally:
        lib.m3p_debug_attn_variant(0)
    assert torch.equal(outs[2], outs[0])                                 # same arithmetic, different schedule
    assert torch.equal(outs[1][:, d:], outs[0][:, d:])                   # dK, dV: phase A is the same code in all three
    assert rel_l2(outs[1][:, :d], outs[0][:, :d]) < 2e-3                 # dQ: from recomputed scores against from the handed-over dS
