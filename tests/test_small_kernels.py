"""Parity of the row / reduction / optimizer kernels against the oracle (GPU)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.util import randn_bf16, randn_f32, rel_l2, max_abs

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def test_cast_and_transpose():
    from m3p_amd import ops
    x, xc = randn_f32((37, 64), 1)
    assert torch.equal(ops.cast_bf16(x).cpu(), xc.to(BF16))
    a, ac = randn_bf16((130, 200), 2)
    dst = torch.zeros((200, 136), dtype=BF16, device='cuda')
    ops.transpose_bf16(a, dst)
    assert torch.equal(dst[:, :130].cpu(), ac.to(BF16).t())
    assert bool((dst[:, 130:] == 0).all())


def test_gather_scatter_colsum():
    from m3p_amd import ops
    src, srcc = randn_bf16((50, 128), 1)
    idx = torch.tensor([3, 49, 0, 17], dtype=torch.int32, device='cuda')
    out = ops.gather_rows(src, idx, 4, 128)
    assert torch.equal(out.cpu(), srcc.to(BF16)[idx.cpu().long()])
    dst = torch.zeros((50, 128), dtype=BF16, device='cuda')
    ops.scatter_add_rows(out, idx, dst, 4, 128)
    ref = torch.zeros(50, 128); ref[idx.cpu().long()] = srcc[idx.cpu().long()]
    assert torch.equal(dst.float().cpu(), ref)
    x, xc = randn_bf16((333, 1024), 3)
    cs = torch.zeros(1000, device='cuda')
    sc = torch.tensor([0.5], device='cuda')
    ops.colsum(x, 1000, cs, scale=sc)
    assert rel_l2(cs, 0.5 * xc[:, :1000].sum(0)) < 1e-5


@pytest.mark.parametrize('n,V', [(9, 1000), (40, 250002), (3, 64)])
def test_cross_entropy_fwd_bwd(n, V):
    from m3p_amd import ops
    ld = (V + 63) // 64 * 64
    logits, lc = randn_bf16((n, ld), 1, 3.0)
    y = torch.from_numpy(np.random.RandomState(2).randint(0, V, size=n)).long()
    x = lc[:, :V].clone().requires_grad_(True)
    ref = F.cross_entropy(x, y, reduction='mean')
    ref.backward()
    loss_sum, row_loss = ops.ce_fwd_bwd(logits, V, y.cuda(), 1.0 / n, 1.0 / n)
    assert abs(float(loss_sum) - float(ref)) < 2e-4 * max(1.0, float(ref))
    assert rel_l2(row_loss, F.cross_entropy(lc[:, :V], y, reduction='none')) < 1e-5
    assert rel_l2(logits[:, :V].float(), x.grad) < 5e-3      # bf16 gradient storage
    assert bool((logits[:, V:] == 0).all())


@pytest.mark.parametrize('n,V', [(9, 1000), (70, 250002), (33, 4100), (1, 64)])
def test_cross_entropy_with_fused_bias_gradient(n, V):
    """The three-launch form (row statistics; gradient tiles that also sum their columns; reduction of the partial sums):
    same loss and gradient as the one-kernel form, and column sums equal to the sum of the rounded gradient rows."""
    from m3p_amd import ops
    ld = (V + 255) // 256 * 256
    logits, lc = randn_bf16((n, ld), 1, 3.0)
    twin = logits.clone()
    y = torch.from_numpy(np.random.RandomState(2).randint(0, V, size=n)).long()
    loss_sum, row_loss, cs = ops.ce_fwd_bwd_colsum(logits, V, y.cuda(), 1.0 / n, 1.0 / n)
    loss_ref, row_ref = ops.ce_fwd_bwd(twin, V, y.cuda(), 1.0 / n, 1.0 / n)
    assert abs(float(loss_sum) - float(loss_ref)) < 1e-5 * max(1.0, float(loss_ref))
    assert rel_l2(row_loss, row_ref) < 1e-6
    assert torch.equal(logits, twin)                      # the same gradient bits, pad columns zero
    assert bool((logits[:, V:] == 0).all())
    ref = F.cross_entropy(lc[:, :V], y, reduction='mean')
    assert abs(float(loss_sum) - float(ref)) < 2e-4 * max(1.0, float(ref))
    assert cs.shape == (ld,) and rel_l2(cs, logits.float().sum(0)) < 1e-5 and float(cs[V:].abs().max()) == 0.0


def test_adam_step_over_ranges_in_one_launch():
    """m3p_adam_step_ranges (every piece of an optimizer step in one launch: the shards of a data-parallel rank, the lazily
    zeroed vocabulary range beside the rest) against m3p_adam_step piece by piece: bit-identical parameters, moments, bf16
    copies; per-piece step sizes and zero flags honoured; what lies between the pieces untouched; more pieces than one
    launch's descriptor holds."""
    from m3p_amd import ops
    n = 1 << 20
    gen = torch.Generator(device='cuda').manual_seed(3)
    bounds = [0, 64, 4096, 4096 + 300 * 64, 400_000, 400_000, 700_000, 790_000]        # (disjoint pieces, one of them empty)
    pieces = [(a, b, 1e-3 * (k + 1), k % 3 != 1) for k, (a, b) in enumerate(zip(bounds[::2], bounds[1::2]))]
    pieces += [(800_000 + 512 * k, 800_000 + 512 * k + 256, 2e-3, True) for k in range(40)]        # > 32 pieces: two launches
    state = [torch.randn(n, device='cuda', generator=gen) for _ in range(3)] + [torch.rand(n, device='cuda', generator=gen)]
    gn = torch.full((1,), 1e4, dtype=torch.float64, device='cuda')
    outs = []
    for form in ('pieces', 'ranges'):
        p, g, m, v = (t.clone() for t in state)
        w16 = torch.zeros(n, dtype=BF16, device='cuda')
        if form == 'ranges':
            ops.adam_step_ranges(p, g, m, v, w16, pieces, 1e-2, 0.9, 0.98, 1e-8, 0.01, gnorm_sq=gn, max_norm=5.0, grad_scale=0.5)
        else:
            for a, b, st, z in pieces:
                if b > a:
                    ops.adam_step(p[a:b], g[a:b], m[a:b], v[a:b], w16[a:b], 1e-2, 0.9, 0.98, 1e-8, 0.01, st, gnorm_sq=gn, max_norm=5.0,
                                  grad_scale=0.5, zero_grad=z)
        torch.cuda.synchronize()
        outs.append((p, g, m, v, w16))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    p, g, m, v, w16 = outs[1]
    assert torch.equal(p[64:4096], state[0][64:4096]) and torch.equal(g[790_000:800_000], state[1][790_000:800_000])     # gaps untouched
    assert float(g[0:64].abs().max()) == 0.0 and torch.equal(g[4096:4096 + 300 * 64], state[1][4096:4096 + 300 * 64])   # zero flag per piece
    assert not torch.equal(p[0:64], state[0][0:64])


def test_adam_step_matches_oracle():
    from m3p_amd import ops
    from oracle import ref_cpu as O
    n = 4096 + 64
    p, pc = randn_f32((n,), 1)
    m = torch.zeros(n, device='cuda'); v = torch.zeros(n, device='cuda')
    mc, vc = torch.zeros(n), torch.zeros(n)
    w16 = torch.zeros(n, dtype=BF16, device='cuda')
    gn = torch.zeros(1, dtype=torch.float64, device='cuda')
    for step in range(1, 4):
        g, gc = randn_f32((n,), 10 + step, 3.0)
        gn.zero_()
        ops.sumsq(g, gn)
        assert abs(math.sqrt(float(gn)) - float(gc.double().norm())) < 1e-6 * float(gc.norm())
        lr = 1e-2
        b1, b2 = 0.9, 0.98
        step_size = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
        ops.adam_step(p, g, m, v, w16, lr, b1, b2, 1e-8, 0.01, step_size, gnorm_sq=gn, max_norm=5.0, grad_scale=1.0)
        (gcl,), _ = O.clip_grad_norm([gc], 5.0)
        pc, mc, vc = O.adam_step(pc, gcl, mc, vc, step, lr, b1, b2, 1e-8, 0.01)
        assert rel_l2(p, pc) < 2e-6
        assert rel_l2(m, mc) < 2e-6 and rel_l2(v, vc) < 5e-6   # g*coef rounding enters v squared
        assert torch.equal(w16.cpu(), p.cpu().to(BF16))
        assert float(g.abs().max()) == 0.0    # zero_grad fused


@pytest.mark.parametrize('p_drop', [0.0, 0.1])
@pytest.mark.parametrize('B,T,R,d', [(8, 64, 10, 128), (5, 24, 36, 768), (4, 16, 0, 128)])
def test_embed_assemble_fwd_bwd(B, T, R, d, p_drop):
    from m3p_amd import ops, rng
    from oracle import ref_cpu as O
    V, S = 300, R + T
    rs = np.random.RandomState(0)
    tok = torch.from_numpy(rs.randint(0, V, size=(T, B))).long()
    lens = torch.from_numpy(rs.randint(max(T // 2, 1), T + 1, size=B)).long()
    for b in range(B):
        tok[lens[b]:, b] = 1
    totlen = (lens + R).int()
    emb16, embc = randn_bf16((V, d), 1, 0.5)
    pos, posc = randn_f32((S + 3, d), 2, 0.1)
    w_loc, w_locc = randn_f32((d, 5), 3, 0.3)
    b_loc, b_locc = randn_f32((d,), 4, 0.1)
    g_img, g_imgc = randn_f32((d,), 5, 0.1); g_img += 1; g_imgc += 1
    be_img, be_imgc = randn_f32((d,), 6, 0.1)
    g_emb, g_embc = randn_f32((d,), 7, 0.1); g_emb += 1; g_embc += 1
    be_emb, be_embc = randn_f32((d,), 8, 0.1)
    img_proj = loc = None
    if R > 0:
        img_proj, img_projc = randn_bf16((R * B, d), 9)
        loc, locc = randn_f32((R, B, 5), 10)
    seed_i, seed_e = 111, 222
    h, saved = ops.embed_assemble_fwd(tok.cuda(), emb16, pos, img_proj, loc, w_loc, b_loc, g_img, be_img, g_emb, be_emb,
                                      totlen.cuda(), B, T, R, d, seed_img=seed_i, seed_emb=seed_e, p_drop=p_drop)
    # oracle restatement of transformer.py:901-943 on the same (bf16-rounded) inputs
    leaves = {k: t.clone().requires_grad_(True) for k, t in dict(
        emb=embc, pos=posc, w_loc=w_locc, b_loc=b_locc, g_img=g_imgc, be_img=be_imgc, g_emb=g_embc, be_emb=be_embc).items()}
    keep_i = keep_e = None
    if p_drop > 0:
        keep_e = torch.from_numpy(rng.keep_mask(B * S * d, seed_e, p_drop, (B, S, d))).float()
        if R > 0:
            keep_i = torch.from_numpy(rng.keep_mask(R * B * d, seed_i, p_drop, (R, B, d))).float().transpose(0, 1)
    tokemb = F.embedding(tok.t(), leaves['emb'])
    if R > 0:
        ip = img_projc.clone().requires_grad_(True)
        e = ip.view(R, B, d) + F.linear(locc, leaves['w_loc'], leaves['b_loc'])
        im = O.layer_norm(e, leaves['g_img'], leaves['be_img']).transpose(0, 1)
        im = O._drop(im, keep_i, p_drop)
        z = torch.cat([im, tokemb], dim=1)
    else:
        z = tokemb
    mask = (torch.arange(S)[None, :] < totlen[:, None].long()).float()
    z = (z + leaves['pos'][:S][None]) * mask[..., None]
    href = O._drop(O.layer_norm(z, leaves['g_emb'], leaves['be_emb']), keep_e, p_drop)
    assert rel_l2(h.float().view(B, S, d), href) < 6e-3
    # backward
    dh, dhc = randn_bf16((B * S, d), 20)
    grads = {k: torch.zeros(s, device='cuda') for k, s in dict(
        d_g_emb=(d,), d_be_emb=(d,), d_pos=(S + 3, d), d_emb=(V, d), d_g_img=(d,), d_be_img=(d,), d_b_img=(d,),
        d_b_loc=(d,), d_w_loc=(d, 5)).items()}
    de = ops.embed_assemble_bwd(dh, saved, g_emb, g_img, tok.cuda(), totlen.cuda(), loc, grads, B, T, R, d, 1,
                                seed_img=seed_i, seed_emb=seed_e, p_drop=p_drop)
    href.backward(dhc.view(B, S, d))
    tol = 1.5e-2
    assert rel_l2(grads['d_g_emb'], leaves['g_emb'].grad) < tol
    assert rel_l2(grads['d_be_emb'], leaves['be_emb'].grad) < tol
    assert rel_l2(grads['d_pos'], leaves['pos'].grad) < tol
    ge = leaves['emb'].grad.clone(); ge[1] = 0          # padding_idx row receives no gradient
    assert rel_l2(grads['d_emb'], ge) < tol
    if R > 0:
        assert rel_l2(grads['d_g_img'], leaves['g_img'].grad) < tol
        assert rel_l2(grads['d_be_img'], leaves['be_img'].grad) < tol
        assert rel_l2(grads['d_b_loc'], leaves['b_loc'].grad) < tol
        assert rel_l2(grads['d_b_img'], leaves['b_loc'].grad) < tol
        assert rel_l2(grads['d_w_loc'], leaves['w_loc'].grad) < tol
        assert rel_l2(de.float(), ip.grad) < tol


def test_gelu_fwd_and_batched_transpose():
    from m3p_amd import ops
    from oracle import ref_cpu as O
    u, uc = randn_bf16((1000, 64), 1, 2.0)
    h = ops.gelu_fwd(u)
    assert rel_l2(h.float(), O.gelu_erf(uc)) < 4e-3
    # same pass with the derivative written over u (what the training path saves for backward)
    u2 = u.clone()
    h2 = ops.gelu_fwd(u2, grad_inplace=True)
    assert torch.equal(h2, h)
    x = uc.double().requires_grad_(True)
    (0.5 * x * (1 + torch.erf(x / 2 ** 0.5))).sum().backward()
    assert rel_l2(u2.float(), x.grad) < 4e-3
    a, ac = randn_bf16((130, 200), 2)
    b, bc = randn_bf16((64, 64), 3)
    da = torch.zeros((200, 130), dtype=BF16, device='cuda'); db = torch.zeros((64, 64), dtype=BF16, device='cuda')
    desc = torch.tensor([[a.data_ptr(), da.data_ptr(), 130, 200, 200, 130], [b.data_ptr(), db.data_ptr(), 64, 64, 64, 64]],
                        dtype=torch.int64, device='cuda')
    ops.transpose_batch(desc, 2, 12)
    assert torch.equal(da.cpu(), ac.to(BF16).t()) and torch.equal(db.cpu(), bc.to(BF16).t())


@pytest.mark.parametrize('B,S,d', [(6, 5, 128), (256, 3, 768)])
def test_itm_head_fwd_bwd(B, S, d):
    """BertPooler + seq_relationship (GEMMs + csrc/itm.hip glue) against fp32 torch on the same bf16 inputs."""
    from m3p_amd import ops
    hid, hc = randn_bf16((B, S, d), 1)                      # (B, S, d): the head reads position 0 of each sequence
    W16, W1c = randn_bf16((d, d), 2, 0.05)                  # the GEMMs read the bf16 working copy of the weight
    b1, b1c = randn_f32((d,), 3, 0.1)
    w2, w2c = randn_f32((d,), 4, 0.1)
    b2, b2c = randn_f32((1,), 5)
    first = hid[:, 0]
    h16, pooled, scores = ops.itm_head_fwd(first, W16, b1, w2, b2)
    x = hc[:, 0].clone().requires_grad_(True)
    P = [t.clone().requires_grad_(True) for t in (W1c, b1c, w2c, b2c)]
    ref_pooled = torch.tanh(F.linear(x, P[0], P[1]))
    ref = F.linear(ref_pooled, P[2].view(1, -1), P[3]).view(-1)
    assert rel_l2(pooled, ref_pooled.detach()) < 4e-3 and rel_l2(scores, ref.detach()) < 4e-3   # bf16 pre-activation
    ds, dsc = randn_f32((B,), 6)
    ref.backward(dsc)
    db1 = torch.zeros(d, device='cuda'); dw2 = torch.zeros(d, device='cuda'); db2 = torch.zeros(1, device='cuda')
    dW1 = torch.zeros((d, d), device='cuda')
    dh = ops.itm_head_bwd(ds, h16, pooled, W16, w2, dW1, db1, dw2, db2)
    assert rel_l2(dh.float(), x.grad) < 8e-3
    assert rel_l2(db1, P[1].grad) < 6e-3 and rel_l2(dw2, P[2].grad) < 6e-3 and rel_l2(db2, P[3].grad) < 1e-5
    assert rel_l2(dW1, P[0].grad) < 8e-3


def test_gelu_bwd_and_mse_kernels():
    from m3p_amd import ops
    dy, dyc = randn_bf16((77, 128), 1)
    u, uc = randn_bf16((77, 128), 2, 2.0)
    du = ops.gelu_bwd(dy, u)
    x = uc.double().requires_grad_(True)
    (0.5 * x * (1 + torch.erf(x / 2 ** 0.5))).backward(dyc.double())
    assert rel_l2(du.float(), x.grad) < 4e-3
    pred, pc = randn_bf16((23, 2048), 3)
    tgt, tc = randn_f32((23, 2048), 4)
    sq, dpred = ops.mse_fwd_bwd(pred, tgt, 1.0 / (23 * 2048))
    p = pc.clone().requires_grad_(True)
    ref = F.mse_loss(p, tc)
    ref.backward()
    assert abs(float(sq) / (23 * 2048) - float(ref)) < 1e-5 * float(ref)
    assert rel_l2(dpred.float(), p.grad) < 4e-3


# ---- host-glue kernels (csrc/glue.hip): each against the chain of tensor ops it replaces ----
def test_seq_masks_and_mask_to_rows():
    from m3p_amd import ops
    B, S, T, R, d = 7, 164, 128, 36, 64
    g = torch.Generator().manual_seed(3)
    lens = torch.randint(T // 2, T + 1, (B,), generator=g)
    limg = torch.full((B,), R, dtype=torch.int64)
    tot, rm = ops.seq_masks(lens.cuda(), limg.cuda(), B, S)
    assert torch.equal(tot.cpu().long(), lens + limg)
    assert torch.equal(rm.cpu().view(B, S).bool(), torch.arange(S)[None, :] < (lens + limg)[:, None])
    tot1, rm1 = ops.seq_masks(lens.cuda(), None, B, T)
    assert torch.equal(tot1.cpu().long(), lens) and int(rm1.sum()) == int(lens.sum())
    # rows of the True entries of a (T, B) mask under the (T, B, d) view out[R:] of a [B, S, d] buffer
    for T2, B2 in ((128, 7), (128, 1024), (5, 3)):
        pm = torch.rand(T2, B2, generator=g) < 0.15
        n = int(pm.sum())
        s0, s1, soff = d, S * d, R * d                         # element strides of encoder_outputs[R:]
        rows = ops.mask_to_rows(pm.cuda(), B2, s0, s1, soff, d, n)
        t_idx, b_idx = pm.nonzero(as_tuple=True)               # (t, b) order
        assert torch.equal(rows.cpu().long(), (soff + t_idx * s0 + b_idx * s1) // d)
    assert ops.mask_to_rows(torch.zeros(4, 4, dtype=torch.bool, device='cuda'), 4, 1, 1, 0, 1, 0).numel() == 0


def test_strided_cast_and_device_scalar_scaling():
    from m3p_amd import ops
    img, imgc = randn_f32((5, 36, 2048), 4)                    # (n, R, 2048) as the collate emits it
    out = ops.cast_rows_bf16(img.transpose(0, 1))              # the (R, n, 2048) view the model is handed
    assert torch.equal(out.cpu(), imgc.transpose(0, 1).contiguous().view(-1, 2048).to(BF16))
    gsc = torch.tensor([0.37], device='cuda')
    x, xc = randn_bf16((33, 64), 5)
    assert torch.equal(ops.scale_bf16_dev(x, gsc).cpu(), (xc.to(BF16).float() * gsc.cpu()).to(BF16))
    y, yc = randn_f32((33, 64), 6)
    assert torch.equal(ops.scale_bf16_dev(y, gsc).cpu(), (yc * gsc.cpu()).to(BF16))
    dst, dstc = randn_f32((1001,), 7)
    src, srcc = randn_f32((1001,), 8)
    ops.axpy_dev(dst, src, gsc)
    assert rel_l2(dst, dstc + 0.37 * srcc) < 1e-6


@pytest.mark.parametrize('w_ce,w_bce,G,n', [(0.0, 1.0, 128, 2), (1.0, 1.0, 24, 4), (0.5, 0.0, 3, 5), (1.0, 2.0, 700, 2)])
def test_itm_loss_kernel_vs_torch(w_ce, w_bce, G, n):
    from m3p_amd import functional as Fn
    g = torch.Generator().manual_seed(G)
    sc = (torch.randn(G * n, 1, generator=g) * 3).requires_grad_(True)
    pos = torch.randint(0, n, (G,), generator=g)
    ref = w_ce * F.cross_entropy(sc.view(-1, n), pos) + \
        w_bce * F.binary_cross_entropy_with_logits(sc.view(-1), F.one_hot(pos, n).float().view(-1))
    ref.backward()
    scg = sc.detach().cuda().requires_grad_(True)
    loss = Fn.ItmLossFn.apply(scg, pos.cuda(), n, w_ce, w_bce)
    (2.0 * loss).backward()
    assert abs(float(loss) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    assert rel_l2(scg.grad, 2.0 * sc.grad) < 1e-5


def test_gradient_sink_equals_autograd_views():
    """The heads leave their rows' gradients in the encoder pass's GradSink instead of returning zero-filled views of the
    whole output: same parameter gradients as the plain autograd route (sink detached), and a consumer outside the
    protocol (a plain tensor op on the output) still adds up."""
    from m3p_amd import synth, functional as Fn
    from m3p_amd.model.transformer import TransformerModel
    P = synth.model_params(128, 4, 2, 1000)
    grads = []
    for use_sink in (True, False):
        torch.manual_seed(0)
        m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True)
        m.load_state_dict(synth.golden_state_dict(synth.hot_param_shapes(P)), strict=False)
        m = m.cuda().train()
        m.dropout = m.attention_dropout = 0.0
        b = synth.make_batch(24, 10, 8, 1000, 4, seed=5, ragged=True)
        out = m('jointfwd', x=b['x'].cuda(), lengths=b['lengths'].cuda(), x_img=b['x_img'].cuda(),
                lengths_img=b['lengths_img'].cuda(), causal=False, langs=None, image_loc=b['image_loc'].cuda(), refine_image=False)
        if not use_sink:
            out._base._m3p_sink = None
        _, mlm = m('predict', tensor=out[10:], pred_mask=b['pred_mask'].cuda(), y=b['x_labels'][b['pred_mask']].cuda(), get_scores=False)
        rel = m('predict', tensor=out.transpose(0, 1), is_relation=True)
        extra = out.float().pow(2).mean()                     # outside the sink protocol
        (mlm + rel.float().mean() + extra).backward()
        torch.cuda.synchronize()
        grads.append(m.arena().grad.clone())
    assert rel_l2(grads[0], grads[1]) < 2e-3, rel_l2(grads[0], grads[1])


def test_batched_transpose_fast_and_ragged_tiles():
    from m3p_amd import ops
    mats = []
    for k, (r, c) in enumerate([(768, 2304), (128, 192), (130, 200), (64, 64), (3072, 768)]):
        a, ac = randn_bf16((r, c), 10 + k)
        dst = torch.zeros((c, (r + 7) // 8 * 8), dtype=BF16, device='cuda')
        mats.append((a, ac, dst))
    rows = [[a.data_ptr(), dst.data_ptr(), a.shape[0], a.shape[1], a.stride(0), dst.stride(0)] for a, _, dst in mats]
    mt = max(((a.shape[0] + 63) // 64) * ((a.shape[1] + 63) // 64) for a, _, _ in mats)
    ops.transpose_batch(torch.tensor(rows, dtype=torch.int64, device='cuda'), len(rows), mt)
    for a, ac, dst in mats:
        assert torch.equal(dst[:, :a.shape[0]].cpu(), ac.to(BF16).t()), tuple(a.shape)


@pytest.mark.gpu
def test_sumsq_over_ranges_in_one_launch():
    """m3p_sumsq_ranges_f32 (the gradient norm of a rank's shards under the sharded data-parallel step) against the
    piecewise sums, including an empty piece, a tiny one and more pieces than one launch's descriptor holds."""
    from m3p_amd import ops
    g = torch.Generator(device='cuda').manual_seed(11)
    buf = torch.randn(3_000_000, device='cuda', generator=g)
    cuts = [(0, 4), (8, 8), (1024, 500_000), (500_004, 500_008), (1_000_000, 2_999_996)]
    cuts += [(2_000_000 + 1024 * i, 2_000_000 + 1024 * i + 512) for i in range(40)]
    out = torch.zeros(1, dtype=torch.float64, device='cuda')
    ops.sumsq_ranges(buf, cuts, out)
    ref = sum(float(buf[a:b].double().pow(2).sum()) for a, b in cuts)
    assert abs(float(out) - ref) <= 1e-6 * ref
