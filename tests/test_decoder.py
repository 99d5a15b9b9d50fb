"""Causal decoder at inference time: crossfwd(causal=True, src_enc=...) with and without the key / value
cache, word scores, generate() and generate_beam() - transformer.py:970-1114, :149-210, :1216-1561.

Golden vectors (tests/golden/decoder.npz) come from the reference's own TransformerModel(is_encoder=False) on the
deterministic cases of m3p_amd.synth.DECODER_CASES (oracle/gen_goldens.py decoder).  CPU tests pin the oracle restatement
and the search loops (driven by the oracle's step function) on them; the GPU tests run the HIP path."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from m3p_amd import synth
from oracle import ref_cpu
from tests.util import rel_l2

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'decoder.npz'))
TAGS = list(synth.DECODER_CASES)


def _langs(c, T, bs):
    return None if c['tgt_lang_id'] is None else torch.full((T, bs), c['tgt_lang_id'], dtype=torch.long)


# ------------------------------------------------------------------------------------------------ CPU: oracle + host logic
@pytest.mark.parametrize('tag', TAGS)
def test_oracle_decoder_matches_the_reference(tag):
    c, P, sd, src_enc, src_len, x, lengths = synth.decoder_case(tag)
    T, bs = x.shape
    out = ref_cpu.decoder_crossfwd(sd, c['n_dec_layers'], c['n_heads'], x, lengths, src_enc, src_len, langs=_langs(c, T, bs))
    assert np.abs(out.numpy() - G[tag + '.full']).max() < 2e-5
    # the reference's cached run (4-token prefix, then token by token) gives the same hidden states
    assert np.abs(G[tag + '.incremental'] - G[tag + '.full']).max() < 2e-5
    assert np.abs(ref_cpu.word_scores(sd, out[-1]).numpy() - G[tag + '.scores_last']).max() < 2e-5
    gen, gen_len, margins = ref_cpu.greedy_decode(sd, c['n_dec_layers'], c['n_heads'], src_enc, src_len, c['tgt_lang_id'], c['max_len'])
    assert np.array_equal(gen.numpy(), G[tag + '.greedy']) and np.array_equal(gen_len.numpy(), G[tag + '.greedy_len'])
    assert np.allclose(margins.numpy(), G[tag + '.greedy_margin'], atol=1e-4)
    # the fixtures exercise both ways a sentence ends
    assert (G[tag + '.greedy_len'] < c['max_len']).any()


def _oracle_backed(monkeypatch, c, sd):
    """m3p_amd.decoder's search loops with the oracle as the step function (full recomputation instead of the HIP path)."""
    from m3p_amd import decoder

    def fwd(model, x, lengths, src_enc, src_len, positions, langs, cache):
        cache['slen'] = x.shape[0]
        return ref_cpu.decoder_crossfwd(sd, c['n_dec_layers'], c['n_heads'], x, lengths, src_enc, src_len, positions, langs)[-1:]

    monkeypatch.setattr(decoder, 'decoder_forward', fwd)
    monkeypatch.setattr(decoder, 'word_scores', lambda model, h: ref_cpu.word_scores(sd, h))
    return SimpleNamespace(n_words=c['n_words'], pad_index=synth.PAD, eos_index=synth.EOS, dim=c['emb_dim'],
                           embeddings=SimpleNamespace(weight=torch.zeros(1)))


@pytest.mark.parametrize('tag', TAGS)
def test_search_loops_on_the_oracle_step_reproduce_the_reference(tag, monkeypatch):
    from m3p_amd import decoder
    c, P, sd, src_enc, src_len, x, lengths = synth.decoder_case(tag)
    stub = _oracle_backed(monkeypatch, c, sd)
    gen, gen_len = decoder.generate(stub, src_enc, src_len, c['tgt_lang_id'], max_len=c['max_len'])
    assert np.array_equal(gen.numpy(), G[tag + '.greedy']) and np.array_equal(gen_len.numpy(), G[tag + '.greedy_len'])
    if c['beam_size']:
        for lp, es in ((1.0, False), (0.6, True)):
            dec, tl = decoder.generate_beam(stub, src_enc, src_len, c['tgt_lang_id'], c['beam_size'], lp, es, max_len=c['max_len'])
            key = '%s.beam_lp%.1f_es%d' % (tag, lp, es)
            assert np.array_equal(tl.numpy(), G[key + '_len']), (key, tl.tolist(), G[key + '_len'].tolist())
            assert np.array_equal(dec.numpy(), G[key]), key


def test_beam_hypotheses_keep_the_n_best():
    from m3p_amd.decoder import BeamHypotheses
    h = BeamHypotheses(2, 11, 1.0, False)
    assert not h.is_done(0.0)
    h.add(torch.arange(4), -4.0)            # score -1.0
    h.add(torch.arange(2), -1.0)            # score -0.5
    assert len(h) == 2 and h.worst_score == -1.0
    h.add(torch.arange(5), -10.0)           # -2.0: worse than the worst, ignored
    assert len(h) == 2
    h.add(torch.arange(4), -1.0)            # -0.25: replaces the -1.0
    assert sorted(s for s, _ in h.hyp) == [-0.5, -0.25] and h.worst_score == -0.5
    assert h.is_done(-6.0) and not h.is_done(-4.0)          # -6 / 10 = -0.6 < -0.5 <= -4 / 10
    assert BeamHypotheses(1, 5, 1.0, True).is_done(0.0) is False


def test_decoder_refuses_training_mode():
    from m3p_amd import decoder
    m = SimpleNamespace(training=True)
    with torch.enable_grad(), pytest.raises(NotImplementedError):
        decoder.decoder_forward(m, torch.zeros(2, 1, dtype=torch.long), torch.ones(1, dtype=torch.long))


# ------------------------------------------------------------------------------------------------ GPU: the HIP path
def _hip_model(tag):
    from m3p_amd.model.transformer import TransformerModel
    c, P, sd, src_enc, src_len, x, lengths = synth.decoder_case(tag)
    torch.manual_seed(0)
    m = TransformerModel(P, is_encoder=False, with_output=True, is_crossModal=True).cuda()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and not [k for k in missing if k.split('.')[0] in
                                   ('attentions', 'encoder_attn', 'ffns', 'layer_norm1', 'layer_norm15', 'layer_norm2')]
    m.eval()
    return m, c, sd, src_enc.cuda(), src_len.cuda(), x.cuda(), lengths.cuda()


@pytest.mark.gpu
@pytest.mark.parametrize('tag', TAGS)
def test_decoder_forward_and_cache_vs_reference(tag):
    m, c, sd, src_enc, src_len, x, lengths = _hip_model(tag)
    T, bs = x.shape
    langs = _langs(c, T, bs)
    langs = None if langs is None else langs.cuda()
    with torch.no_grad():
        full = m('crossfwd', x=x, lengths=lengths, causal=True, src_enc=src_enc, src_len=src_len, langs=langs)
        assert full.shape == (T, bs, c['emb_dim'])
        assert rel_l2(full.float().cpu(), torch.from_numpy(G[tag + '.full'])) < 2e-2
        # incrementally through the cache, as the golden run: the same numbers as the one-shot run (same kernels, same
        # operands - only the GEMM row counts differ)
        cache = {'slen': 0}
        pieces = [m('crossfwd', x=x[:4], lengths=lengths.clamp(max=4), causal=True, src_enc=src_enc, src_len=src_len,
                    langs=None if langs is None else langs[:4], cache=cache)]
        for t in range(5, T + 1):
            pieces.append(m('crossfwd', x=x[:t], lengths=lengths.clamp(max=t), causal=True, src_enc=src_enc, src_len=src_len,
                            langs=None if langs is None else langs[:t], cache=cache))
            assert pieces[-1].shape == (1, bs, c['emb_dim']) and cache['slen'] == t
        inc = torch.cat(pieces, 0)
        assert rel_l2(inc.float().cpu(), torch.from_numpy(G[tag + '.incremental'])) < 2e-2
        assert rel_l2(inc.float(), full.float()) < 5e-3
        scores = m.pred_layer.get_scores(full[-1])
        assert scores.dtype == torch.float32 and scores.shape == (bs, c['n_words'])
        assert rel_l2(scores.cpu(), torch.from_numpy(G[tag + '.scores_last'])) < 2e-2
        # without a source: the plain causal language-model stack (no encoder attention)
        lm = m('crossfwd', x=x, lengths=lengths, causal=True, langs=langs)
        o = ref_cpu.decoder_crossfwd(sd, c['n_dec_layers'], c['n_heads'], x.cpu(), lengths.cpu(), langs=None if langs is None else langs.cpu())
        assert rel_l2(lm.float().cpu(), o) < 2e-2


def _agree_until_near_tie(ours, ref, margins, tol):
    """Token sequences (len, bs) must agree up to (excluding) the first step at which the reference's own top-2 margin
    is below `tol` - from there on bf16 arithmetic may legitimately pick the other word and the continuations differ."""
    n = min(ours.shape[0], ref.shape[0])
    for b in range(ref.shape[1]):
        for t in range(1, n):
            if margins[t - 1, b] < tol:
                break
            assert ours[t, b] == ref[t, b], (b, t, ours[:, b].tolist(), ref[:, b].tolist(), float(margins[t - 1, b]))
    return True


@pytest.mark.gpu
@pytest.mark.parametrize('tag', TAGS)
def test_generate_greedy_vs_reference(tag):
    m, c, sd, src_enc, src_len, x, lengths = _hip_model(tag)
    with torch.no_grad():
        gen, gen_len = m.generate(src_enc, src_len, c['tgt_lang_id'], max_len=c['max_len'])
    ref, ref_len, margins = G[tag + '.greedy'], G[tag + '.greedy_len'], G[tag + '.greedy_margin']
    gen, gen_len = gen.cpu().numpy(), gen_len.cpu().numpy()
    assert gen.shape[1] == ref.shape[1] and (gen[0] == synth.EOS).all()
    assert ((gen == synth.EOS).sum(0) == 2).all()
    _agree_until_near_tie(gen, ref, margins, tol=0.02)
    exact = [b for b in range(ref.shape[1]) if margins[:, b].min() >= 0.02]      # (score differences seen: ~2e-3)
    assert exact, 'the fixture should hold at least one sentence without a near-tie'
    for b in exact:
        assert gen_len[b] == ref_len[b] and np.array_equal(gen[:gen_len[b], b], ref[:ref_len[b], b])
    # sampling: same shapes and invariants, reproducible under a seed
    torch.manual_seed(5)
    with torch.no_grad():
        s1, l1 = m.generate(src_enc, src_len, c['tgt_lang_id'], max_len=c['max_len'], sample_temperature=0.7)
        torch.manual_seed(5)
        s2, l2 = m.generate(src_enc, src_len, c['tgt_lang_id'], max_len=c['max_len'], sample_temperature=0.7)
    assert torch.equal(s1, s2) and torch.equal(l1, l2) and int((s1 == synth.EOS).sum()) == 2 * s1.shape[1]


@pytest.mark.gpu
@pytest.mark.parametrize('tag', [t for t in TAGS if synth.DECODER_CASES[t]['beam_size']])
def test_generate_beam_vs_reference(tag):
    m, c, sd, src_enc, src_len, x, lengths = _hip_model(tag)
    n_equal = n_total = 0
    for lp, es in ((1.0, False), (0.6, True)):
        with torch.no_grad():
            dec, tl = m.generate_beam(src_enc, src_len, c['tgt_lang_id'], c['beam_size'], lp, es, max_len=c['max_len'])
        key = '%s.beam_lp%.1f_es%d' % (tag, lp, es)
        ref, ref_len = G[key], G[key + '_len']
        dec, tl = dec.cpu().numpy(), tl.cpu().numpy()
        assert dec.shape[1] == ref.shape[1] and ((dec == synth.EOS).sum(0) == 2).all() and (dec[0] == synth.EOS).all()
        for b in range(ref.shape[1]):
            n_total += 1
            n_equal += int(tl[b] == ref_len[b] and np.array_equal(dec[:tl[b], b], ref[:ref_len[b], b]))
    # hypotheses whose cumulated log-probabilities are closer than bf16 resolution may swap; most must be identical
    assert n_equal >= n_total - 1, (n_equal, n_total)


@pytest.mark.gpu
def test_attn_query_kernel_vs_torch():
    from m3p_amd import ops
    g = torch.Generator(device='cuda').manual_seed(3)
    for B, Tq, H, dh, Lk, causal, pos0 in ((3, 1, 4, 32, 37, True, 36), (2, 5, 12, 64, 9, True, 4), (4, 3, 2, 64, 200, False, 0),
                                            (1, 1, 16, 64, 700, False, 0)):
        d = H * dh
        q = torch.randn(B * Tq, d, device='cuda', generator=g).to(torch.bfloat16)
        kv = torch.randn(B, Lk + 3, 2 * d, device='cuda', generator=g).to(torch.bfloat16)
        klen = None if causal else torch.randint(1, Lk + 1, (B,), device='cuda', generator=g).to(torch.int32)
        ctx = ops.attn_query_fwd(q, kv, klen, B, Tq, H, dh, Lk, causal=causal, pos0=pos0)
        qf = q.float().view(B, Tq, H, dh).transpose(1, 2)
        kf = kv[:, :Lk, :d].float().reshape(B, Lk, H, dh).transpose(1, 2)
        vf = kv[:, :Lk, d:].float().reshape(B, Lk, H, dh).transpose(1, 2)
        s = qf @ kf.transpose(2, 3)
        j = torch.arange(Lk, device='cuda')
        if causal:
            ok = j[None, :] <= (pos0 + torch.arange(Tq, device='cuda'))[:, None]
            s = s.masked_fill(~ok[None, None], float('-inf'))
        else:
            s = s.masked_fill(~(j[None, :] < klen[:, None])[:, None, None, :], float('-inf'))
        ref = (torch.softmax(s, -1) @ vf).transpose(1, 2).reshape(B * Tq, d)
        assert rel_l2(ctx.float(), ref) < 5e-3, (B, Tq, H, dh, Lk)


def test_language_ids_need_a_multilingual_model():
    from m3p_amd.model.transformer import TransformerModel
    P = synth.model_params(128, 4, 1, 100)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True)
    x = torch.full((6, 2), 5, dtype=torch.long)
    with pytest.raises(AssertionError, match='n_langs'):
        m('crossfwd', stream_='text', x=x, lengths=torch.tensor([6, 4]), langs=torch.zeros_like(x), causal=False)


@pytest.mark.gpu
@pytest.mark.parametrize('B,Tq,H,dh,Lk,causal,p', [(3, 9, 4, 32, 9, True, 0.0), (2, 17, 12, 64, 17, True, 0.1),
                                                   (4, 6, 2, 64, 50, False, 0.1), (2, 33, 4, 32, 137, False, 0.0)])
def test_attn_rows_training_kernels_vs_autograd(B, Tq, H, dh, Lk, causal, p):
    """Forward with dropout (keep mask regenerated by the RNG twin) + log-sum-exp, and the backward (dq of the unscaled
    projection, dk / dv accumulated in fp32) against torch autograd on the same bf16 operands."""
    from m3p_amd import ops, rng
    d = H * dh
    g = torch.Generator(device='cuda').manual_seed(7)
    qscale = 1.0 / np.sqrt(dh)
    q = (torch.randn(B * Tq, d, device='cuda', generator=g) * qscale).to(torch.bfloat16)       # the scaled projection output
    kv = torch.randn(B, Lk, 2 * d, device='cuda', generator=g).to(torch.bfloat16)
    klen = None if causal else torch.randint(1, Lk + 1, (B,), device='cuda', generator=g).to(torch.int32)
    dctx = torch.randn(B * Tq, d, device='cuda', generator=g).to(torch.bfloat16)
    seed = 4242
    ctx, lse = ops.attn_rows_fwd(q, kv, klen, B, Tq, H, dh, Lk, causal=causal, seed=seed, p_drop=p)
    keep = torch.from_numpy(rng.keep_mask(B * H * Tq * Lk, seed, p, (B, H, Tq, Lk))).cuda() if p > 0 else None
    qf = q.float().view(B, Tq, H, dh).transpose(1, 2).requires_grad_(True)
    kf = kv[:, :, :d].float().reshape(B, Lk, H, dh).transpose(1, 2).requires_grad_(True)
    vf = kv[:, :, d:].float().reshape(B, Lk, H, dh).transpose(1, 2).requires_grad_(True)
    s = qf @ kf.transpose(2, 3)
    j = torch.arange(Lk, device='cuda')
    if causal:
        s = s.masked_fill(~(j[None, :] <= torch.arange(Tq, device='cuda')[:, None])[None, None], float('-inf'))
    else:
        s = s.masked_fill(~(j[None, :] < klen[:, None])[:, None, None, :], float('-inf'))
    pr = torch.softmax(s, -1)
    if keep is not None:
        pr = pr * keep / (1 - p)
    ref = (pr @ vf).transpose(1, 2).reshape(B * Tq, d)
    assert rel_l2(ctx.float(), ref) < 5e-3
    assert rel_l2(lse, torch.logsumexp(s, -1)) < 1e-4
    ref.backward(dctx.float())
    dq, dkv = ops.attn_rows_bwd(q, kv, klen, dctx, lse, B, Tq, H, dh, Lk, qscale, causal=causal, seed=seed, p_drop=p)
    assert rel_l2(dq.float(), (qf.grad * qscale).transpose(1, 2).reshape(B * Tq, d)) < 1e-2
    assert rel_l2(dkv[:, :, :d], kf.grad.transpose(1, 2).reshape(B, Lk, d)) < 1e-2
    assert rel_l2(dkv[:, :, d:], vf.grad.transpose(1, 2).reshape(B, Lk, d)) < 1e-2


def _mt_model(extra=None):
    from m3p_amd.model.transformer import TransformerModel
    P, sd, x1, len1, x2, len2 = synth.mt_case()
    for k, v in (extra or {}).items():
        setattr(P, k, v)
    torch.manual_seed(0)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True).cuda()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and not [k for k in missing if k.startswith(('encoder_attn', 'layer_norm15', 'cross_lang'))]
    return m, P, sd, x1, len1, x2, len2


def test_oracle_mt_step_matches_the_reference():
    """CPU: the restatement of the translation step (encoder pass with language ids, teacher-forced causal pass with
    attention over it, next-word loss) against the reference's recorded outputs."""
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'mt_step.npz'))
    P, sd, x1, len1, x2, len2 = synth.mt_case()
    pred_mask, y = synth.mt_targets(x2, len2)
    enc = ref_cpu.crossfwd_text(sd, P.n_layers, P.n_heads, x1, len1, langs=x1.clone().fill_(0)).transpose(0, 1)
    dec = ref_cpu.decoder_crossfwd(sd, P.n_layers, P.n_heads, x2, len2, enc, len1, langs=x2.clone().fill_(1))
    assert np.abs(enc.numpy() - g['enc1']).max() < 2e-5 and np.abs(dec.numpy() - g['dec2']).max() < 2e-5
    loss = ref_cpu.predict_mlm(sd, dec, pred_mask, y)
    loss = loss[1] if isinstance(loss, tuple) else loss
    assert abs(float(loss) - float(g['loss'])) < 1e-5


@pytest.mark.gpu
def test_mt_step_forward_backward_vs_reference():
    """The translation step on the HIP path (DecoderFn: causal self-attention, attention over the encoder pass of the same
    model, gradients through both passes) against the reference's loss and gradients (tests/golden/mt_step.npz)."""
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'mt_step.npz'))
    m, P, sd, x1, len1, x2, len2 = _mt_model()
    assert m.cross_attention_hot and 'encoder_attn.0.q_lin.weight' in m.arena().offsets
    m.train()
    m.arena().zero_grad()
    pred_mask, y = synth.mt_targets(x2, len2)
    enc1 = m('crossfwd', stream_='text', x=x1.cuda(), lengths=len1.cuda(), langs=x1.clone().fill_(0).cuda(), causal=False).transpose(0, 1)
    dec2 = m('crossfwd', stream_='text', x=x2.cuda(), lengths=len2.cuda(), langs=x2.clone().fill_(1).cuda(), causal=True,
             src_enc=enc1, src_len=len1.cuda())
    assert rel_l2(enc1.float(), g['enc1']) < 1e-2 and rel_l2(dec2.float(), g['dec2']) < 1.5e-2
    _, loss = m('predict', tensor=dec2, pred_mask=pred_mask.cuda(), y=y.cuda(), get_scores=False)
    assert abs(float(loss.detach()) - float(g['loss'])) < 5e-3
    loss.backward()
    own = dict(m.named_parameters())
    bad = []
    for k in [k[5:] for k in g.files if k.startswith('grad.')]:
        ref = g['grad.' + k]
        if np.abs(ref).max() < 1e-7:            # (the key biases: exactly zero up to fp32 noise)
            continue
        e = rel_l2(own[k].grad.float(), ref)
        if e > 4e-2:
            bad.append((k, e))
    assert not bad, bad
    ge = own['embeddings.weight'].grad.float()
    assert abs(float(ge.norm()) - float(g['grad_norm.embeddings.weight'])) < 3e-2 * float(g['grad_norm.embeddings.weight'])
    touched = m.arena().touched
    assert {'encoder_attn.1.out_lin.weight', 'layer_norm15.0.bias', 'cross_lang_embeddings.weight'} <= touched


@pytest.mark.gpu
def test_mt_step_trains_and_inference_follows_the_updated_weights():
    """Trainer.mt_step_on_batch: the loss goes down over a few Adam steps with dropout on; generate() afterwards reads the
    UPDATED encoder-attention weights (arena views, not stale copies)."""
    from m3p_amd.trainer import XTrainer
    extra = dict(optimizer='adam_inverse_sqrt,beta1=0.9,beta2=0.98,lr=0.002,warmup_updates=4', clip_grad_norm=5, amp=-1, fp16=False,
                 accumulate_gradients=1, multi_gpu=False, epoch_size=100, cross_mlm_steps=[], cross_mrm_steps=[], langs=['en', 'zh'],
                 cross_mrfr_steps=[], cross_clcm_steps=[], sample_n=2, refine_image=False, batch_size=6, dropout=0.1,
                 attention_dropout=0.1, dump_path='/nonexistent_m3p_dump')
    m, P, sd, x1, len1, x2, len2 = _mt_model(extra)
    m.dropout = m.attention_dropout = 0.1
    tr = XTrainer(m, {}, P)
    losses = []
    for _ in range(12):
        losses.append(float(tr.mt_step_on_batch(x1, len1, x2, len2, 'en', 'zh', 1.0)))
        tr.iter()
    assert np.isfinite(losses).all() and losses[-1] < losses[0] - 0.5, losses
    assert tr.stats['processed_w'] % int((len2 - 1).sum()) == 0       # (print_stats resets the counters every few iterations)
    m.eval()
    with torch.no_grad():
        enc = m('crossfwd', stream_='text', x=x1.cuda(), lengths=len1.cuda(), langs=x1.clone().fill_(0).cuda(), causal=False).transpose(0, 1)
        dec_inf = m('crossfwd', stream_='text', x=x2.cuda(), lengths=len2.cuda(), langs=x2.clone().fill_(1).cuda(), causal=True,
                    src_enc=enc, src_len=len1.cuda())
    # the same forward through the training kernels with dropout off
    m.train()
    m.dropout = m.attention_dropout = 0.0
    dec_tr = m('crossfwd', stream_='text', x=x2.cuda(), lengths=len2.cuda(), langs=x2.clone().fill_(1).cuda(), causal=True,
               src_enc=enc, src_len=len1.cuda())
    assert rel_l2(dec_inf.float(), dec_tr.float().detach()) < 5e-3
    m.eval()
    with torch.no_grad():
        gen, gen_len = m.generate(enc, len1.cuda(), 1, max_len=12)
    assert gen.shape[1] == 6 and int((gen == synth.EOS).sum()) == 12


def test_oracle_ic_step_matches_the_reference():
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ic_step.npz'))
    P, sd, x_img, loc, img_len, x2, len2 = synth.ic_case()
    R, B = x_img.shape[0], x_img.shape[1]
    enc = ref_cpu.crossfwd_img(sd, P.n_layers, P.n_heads, x_img, img_len, loc, langs=torch.zeros((R, B), dtype=torch.long)).transpose(0, 1)
    dec = ref_cpu.decoder_crossfwd(sd, P.n_layers, P.n_heads, x2, len2, enc, img_len, langs=x2.clone().fill_(0))
    assert np.abs(enc.numpy() - g['enc1']).max() < 2e-5 and np.abs(dec.numpy() - g['dec2']).max() < 2e-5
    pred_mask, y = synth.mt_targets(x2, len2)
    loss = ref_cpu.predict_mlm(sd, dec, pred_mask, y)
    loss = loss[1] if isinstance(loss, tuple) else loss
    assert abs(float(loss) - float(g['loss'])) < 1e-5


@pytest.mark.gpu
def test_ic_step_vs_reference():
    """The captioning step: image-only encoder stream (crossfwd stream_='img': BertImageEmbeddings + language embedding, no
    positions / layer_norm_emb), teacher-forced caption decoding over it - loss and gradients against the reference's
    (tests/golden/ic_step.npz), through Trainer.ic_step_on_batch on the tuple layout the reference's collate emits."""
    from m3p_amd.model.transformer import TransformerModel
    from m3p_amd.trainer import XTrainer
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ic_step.npz'))
    P, sd, x_img, loc, img_len, x2, len2 = synth.ic_case()
    for k, v in dict(optimizer='adam_inverse_sqrt,beta1=0.9,beta2=0.98,lr=0.0001', clip_grad_norm=5, amp=-1, fp16=False,
                     accumulate_gradients=1, multi_gpu=False, epoch_size=100, cross_mlm_steps=[], cross_mrm_steps=[], langs=['en', 'zh'],
                     cross_mrfr_steps=[], cross_clcm_steps=[], sample_n=2, refine_image=False, batch_size=6, ft_lgs=[],
                     dump_path='/nonexistent_m3p_dump').items():
        setattr(P, k, v)
    torch.manual_seed(0)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True).cuda()
    m.load_state_dict(sd, strict=False)
    m.train()
    R, B = x_img.shape[0], x_img.shape[1]
    enc1 = m('crossfwd', stream_='img', x=x_img.cuda(), lengths=img_len.cuda(), langs=torch.zeros((R, B), dtype=torch.long).cuda(),
             causal=False, image_loc=loc.cuda(), refine_image=False).transpose(0, 1)
    assert rel_l2(enc1.float(), g['enc1']) < 1e-2
    m.arena().zero_grad()
    tr = XTrainer(m, {}, P)
    grads = {}
    opt = tr.optimizers['model']
    inner = opt.step

    def step(closure=None):          # look at the gradients the optimizer is about to consume
        torch.cuda.synchronize()
        for k in [k[5:] for k in g.files if k.startswith('grad.')]:
            grads[k] = dict(m.named_parameters())[k].grad.float().cpu().clone()
        return inner(closure)
    opt.step = step
    x1_mask = (torch.arange(R)[None, :] < img_len[:, None]).long()
    loss = tr.ic_step_on_batch(x2, len2, x_img.transpose(0, 1).contiguous(), x1_mask, loc.transpose(0, 1).contiguous(), 'coco', 'img', 1.0)
    assert abs(float(loss) - float(g['loss'])) < 5e-3
    bad = [(k, rel_l2(v, g['grad.' + k])) for k, v in grads.items()]
    bad = [(k, e) for k, e in bad if e > 4e-2]
    assert grads and not bad, bad
    # the clip norm of this step is far below 5: the gradients above are the unclipped ones
    assert opt.grad_norm() < 5


def test_oracle_mt_ic_step_matches_the_reference():
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'mt_ic_step.npz'))
    P, sd, x_src, len_src, x_img, loc, img_len, x2, len2 = synth.mt_ic_case()
    enc = ref_cpu.jointfwd(sd, P.n_layers, P.n_heads, x_src, len_src, x_img, img_len, loc).transpose(0, 1)
    dec = ref_cpu.decoder_crossfwd(sd, P.n_layers, P.n_heads, x2, len2, enc, len_src + img_len, langs=x2.clone().fill_(1))
    assert np.abs(enc.numpy() - g['enc1']).max() < 2e-5 and np.abs(dec.numpy() - g['dec2']).max() < 2e-5
    pred_mask, y = synth.mt_targets(x2, len2)
    loss = ref_cpu.predict_mlm(sd, dec, pred_mask, y)
    loss = loss[1] if isinstance(loss, tuple) else loss
    assert abs(float(loss) - float(g['loss'])) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize('only_text', [False, True])
def test_mt_ic_step_vs_reference(only_text):
    """The multimodal-translation step: jointfwd over (regions | source words) as the encoder pass, the target decoded over all
    R + len_src positions - loss and gradients against the reference's (tests/golden/mt_ic_step.npz) through
    Trainer.mt_ic_step_on_batch on the tuple layout of mt_caption_collate.  With params.mt_only_text the encoder pass is the
    text stream alone, which is mt_step's computation: checked against tests/golden/mt_step.npz."""
    from m3p_amd.model.transformer import TransformerModel
    from m3p_amd.trainer import XTrainer
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'mt_step.npz' if only_text else 'mt_ic_step.npz'))
    P, sd, x_src, len_src, x_img, loc, img_len, x2, len2 = synth.mt_ic_case()
    for k, v in dict(optimizer='adam_inverse_sqrt,beta1=0.9,beta2=0.98,lr=0.0001', clip_grad_norm=5, amp=-1, fp16=False,
                     accumulate_gradients=1, multi_gpu=False, epoch_size=100, cross_mlm_steps=[], cross_mrm_steps=[], langs=['en', 'zh'],
                     cross_mrfr_steps=[], cross_clcm_steps=[], sample_n=2, refine_image=False, batch_size=6, ft_lgs=['en', 'zh'],
                     mt_only_text=only_text, dump_path='/nonexistent_m3p_dump').items():
        setattr(P, k, v)
    torch.manual_seed(0)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True).cuda()
    m.load_state_dict(sd, strict=False)
    m.train()
    tr = XTrainer(m, {}, P)
    grads = {}
    opt = tr.optimizers['model']
    inner = opt.step

    def step(closure=None):
        torch.cuda.synchronize()
        for k in [k[5:] for k in g.files if k.startswith('grad.')]:
            grads[k] = dict(m.named_parameters())[k].grad.float().cpu().clone()
        grads['|embeddings.weight|'] = dict(m.named_parameters())['embeddings.weight'].grad.float().norm().cpu()
        return inner(closure)
    opt.step = step
    R, B = x_img.shape[0], x_img.shape[1]
    x1_mask = torch.ones(B, R, dtype=torch.long)
    loss = tr.mt_ic_step_on_batch(x_src, len_src, x2, len2, x_img.transpose(0, 1).contiguous(), x1_mask,
                                  loc.transpose(0, 1).contiguous(), 'coco', 'img', 1.0)
    assert abs(float(loss) - float(g['loss'])) < 5e-3
    enorm = grads.pop('|embeddings.weight|')
    assert abs(float(enorm) - float(g['grad_norm.embeddings.weight'])) < 4e-2 * float(g['grad_norm.embeddings.weight'])
    bad = [(k, rel_l2(v, g['grad.' + k])) for k, v in grads.items()]
    bad = [(k, e) for k, e in bad if e > 4e-2]
    assert grads and not bad, bad
    assert opt.grad_norm() < 5
    assert tr.stats['processed_s'] == B and tr.stats['processed_w'] == int((len2 - 1).sum())


def _step_params(P, **over):
    base = dict(optimizer='adam_inverse_sqrt,beta1=0.9,beta2=0.98,lr=0.0001', clip_grad_norm=5, amp=-1, fp16=False,
                accumulate_gradients=1, multi_gpu=False, epoch_size=100, cross_mlm_steps=[], cross_mrm_steps=[], langs=['en', 'zh'],
                cross_mrfr_steps=[], cross_clcm_steps=[], sample_n=2, refine_image=False, batch_size=6, ft_lgs=['en', 'zh'],
                group_by_size=False, dump_path='/nonexistent_m3p_dump')
    base.update(over)
    for k, v in base.items():
        setattr(P, k, v)
    return P


def _grads_at_step(m, opt, names):
    """Gradients the optimizer is about to consume, captured by wrapping its step()."""
    got = {}
    inner = opt.step

    def step(closure=None):
        torch.cuda.synchronize()
        named = dict(m.named_parameters())
        for k in names:
            got[k] = named[k].grad.float().cpu().clone()
        return inner(closure)
    opt.step = step
    return got


@pytest.mark.gpu
def test_ntg_step_vs_oracle():
    """Text-to-text generation (xtrainer.py:2596-2645): a (source, target) batch of one language out of data['text'] - the
    translation computation with the same language id on both sides.  Loss and gradients against the oracle's autograd (the
    oracle's encoder / decoder passes are pinned to the reference by tests/golden/mt_step.npz)."""
    from m3p_amd.model.transformer import TransformerModel
    from m3p_amd.trainer import XTrainer
    P, sd, x1, len1, x2, len2 = synth.mt_case()
    _step_params(P, is_ntg=True)

    class Pairs:
        def get_iterator(self, shuffle, group_by_size=False, n_sentences=-1):
            assert shuffle and n_sentences == -1
            return iter([((x1, len1), (x2, len2))])

    torch.manual_seed(0)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True).cuda()
    m.load_state_dict(sd, strict=False)
    tr = XTrainer(m, {'text': {'zh': {'train': Pairs()}}}, P)
    names = ['cross_lang_embeddings.weight', 'attentions.0.q_lin.weight', 'encoder_attn.1.k_lin.weight', 'encoder_attn.0.out_lin.bias',
             'ffns.1.lin1.weight', 'layer_norm15.0.weight', 'pred_layer.proj.bias']
    grads = _grads_at_step(m, tr.optimizers['model'], names)
    loss = tr.ntg_step('zh', None, 1.0)
    # oracle: both sides carry language id 1
    ref = {k: v.clone().requires_grad_(k in names) for k, v in sd.items()}
    enc = ref_cpu.crossfwd_text(ref, P.n_layers, P.n_heads, x1, len1, langs=x1.clone().fill_(1)).transpose(0, 1)
    dec = ref_cpu.decoder_crossfwd(ref, P.n_layers, P.n_heads, x2, len2, enc, len1, langs=x2.clone().fill_(1))
    pred_mask, y = synth.mt_targets(x2, len2)
    o = ref_cpu.predict_mlm(ref, dec, pred_mask, y)
    o = o[1] if isinstance(o, tuple) else o
    o.backward()
    assert abs(float(loss) - float(o.detach())) < 5e-3
    bad = [(k, rel_l2(grads[k], ref[k].grad.numpy())) for k in names]
    assert not [b for b in bad if b[1] > 4e-2], bad
    assert 'NTG-zh' in tr.stats and tr.stats['processed_w'] == int((len2 - 1).sum())
    assert tr.ntg_step('zh', None, 0) is None


@pytest.mark.gpu
def test_slide_step_vs_oracle():
    """Sliding-window matching (xtrainer.py:2649-2698): slide_collate batch -> jointfwd -> relation score -> BCE against 0/1
    labels; loss and gradients against the oracle's autograd, the batch drawn through the DataLoader the is_slide flag selects."""
    from m3p_amd.model.transformer import TransformerModel
    from m3p_amd.trainer import XTrainer
    P, sd, x_src, len_src, x_img, loc, img_len, _, _ = synth.mt_ic_case()
    _step_params(P, is_slide=True, is_generation=False, is_pretrain=False, n_gpu_per_node=1, num_workers=0, batch_size=3)
    R, B = x_img.shape[0], x_img.shape[1]                 # six sequences = three items of two windows each
    labels = [1, 0, 0, 1, 1, 1]

    class Items(torch.utils.data.Dataset):
        def __len__(self):
            return 3

        def __getitem__(self, i):
            cols = [2 * i, 2 * i + 1]
            sents = [x_src[1:int(len_src[c]) - 1, c].numpy() for c in cols]
            return (sents, x_img[:, cols].transpose(0, 1).contiguous(), torch.ones(2, R, dtype=torch.long),
                    loc[:, cols].transpose(0, 1).contiguous(), [10 * c for c in cols], [labels[c] for c in cols])

    torch.manual_seed(0)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True).cuda()
    m.load_state_dict(sd, strict=False)
    tr = XTrainer(m, {'cross_modal': {('slide', 'img'): {'train': Items()}}}, P)
    names = ['pooled_layer.dense.weight', 'seq_relationship.weight', 'attentions.1.v_lin.weight', 'ffns.0.lin2.weight',
             'image_embeddings.image_embeddings.weight', 'layer_norm_emb.bias']
    grads = _grads_at_step(m, tr.optimizers['model'], names)
    order = []
    real_get = tr.get_batch

    def get_batch(*a):                                     # the sampler shuffles items: record which ones came
        b = real_get(*a)
        order.extend(b[1][3])
        return b
    tr.get_batch = get_batch
    loss = tr.slide_step('slide', 'img', 1.0)
    cols = [i // 10 for i in order]
    assert sorted(cols) == list(range(6))
    ref = {k: v.clone().requires_grad_(k in names) for k, v in sd.items()}
    # BOS/EOS framing of batch_sentences: the reference marks both sentence ends with EOS in its streams, the collate uses 0 / 2
    xs = x_src[:, cols].clone()
    xs[0] = 0
    enc = ref_cpu.jointfwd(ref, P.n_layers, P.n_heads, xs, len_src[cols], x_img[:, cols], img_len[cols], loc[:, cols])
    scores = ref_cpu.predict_relation(ref, enc.transpose(0, 1))
    o = torch.nn.functional.binary_cross_entropy_with_logits(scores.view(-1), torch.tensor([float(labels[c]) for c in cols]))
    o.backward()
    assert abs(float(loss) - float(o.detach())) < 5e-3
    bad = [(k, rel_l2(grads[k], ref[k].grad.numpy())) for k in names]
    assert not [b for b in bad if b[1] > 4e-2], bad
    assert 'SLIDE-img' in tr.stats and tr.stats['processed_s'] == 6


def test_oracle_image_stream_with_refiner_matches_the_reference():
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ic_refine.npz'))
    P, sd, x_img, loc, img_len, w = synth.ic_refine_case()
    R, B = x_img.shape[0], x_img.shape[1]
    names = [k[5:] for k in g.files if k.startswith('grad.')]
    ref = {k: v.clone().requires_grad_(k in names) for k, v in sd.items()}
    enc = ref_cpu.crossfwd_img(ref, P.n_layers, P.n_heads, x_img, img_len, loc, langs=torch.ones((R, B), dtype=torch.long), n_refine_layers=2)
    assert np.abs(enc.detach().numpy() - g['enc']).max() < 2e-5
    (enc * w).sum().backward()
    # (the key projection's bias shifts every score of a query alike: its exact gradient is 0, both sides hold rounding noise)
    kbias = [k for k in names if k.endswith('self_attn.linears.1.bias')]
    assert len(kbias) == 2 and all(float(ref[k].grad.abs().max()) < 1e-5 and float(np.abs(g['grad.' + k]).max()) < 1e-5 for k in kbias)
    bad = [(k, rel_l2(ref[k].grad, g['grad.' + k])) for k in names if k not in kbias]
    assert not [b for b in bad if b[1] > 1e-4], bad


@pytest.mark.gpu
def test_image_stream_with_refiner_vs_reference():
    """crossfwd(stream_='img', refine_image=True) - the captioning encoder pass as the reference's default flags run it
    (train_x.py:285): BertImageEmbeddings (+ language embedding) -> dropout -> mask -> AoA refiner -> layers; the encoding and
    the gradients of sum(enc * w) for all 34 refiner parameters and for parameters before and behind it against the
    reference's (tests/golden/ic_refine.npz, eval mode like the jointfwd refiner golden)."""
    from m3p_amd.model.transformer import TransformerModel
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ic_refine.npz'))
    P, sd, x_img, loc, img_len, w = synth.ic_refine_case()
    R, B = x_img.shape[0], x_img.shape[1]
    torch.manual_seed(0)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True).cuda()
    m.load_state_dict(sd, strict=False)
    m.eval()
    m.arena().zero_grad()
    enc = m('crossfwd', stream_='img', x=x_img.cuda(), lengths=img_len.cuda(), langs=torch.ones((R, B), dtype=torch.long).cuda(),
            causal=False, image_loc=loc.cuda(), refine_image=True)
    assert rel_l2(enc.float(), g['enc']) < 1e-2
    (enc.float() * w.cuda()).sum().backward()
    torch.cuda.synchronize()
    named = dict(m.named_parameters())
    names = [k[5:] for k in g.files if k.startswith('grad.')]
    assert sum(k.startswith('refine_embeddings.') for k in names) == 34
    # With N(0, 0.02) weights the attention scores are ~0 and every softmax is nearly uniform: the gradients of the query / key
    # projections are second-order small - 1e-5 against 1.5 for the value projection in the reference's own numbers - and
    # below the rounding of the bf16 rows they are computed from.  Those are held to that scale, the others to 4e-2.
    vnorm = float(np.linalg.norm(g['grad.refine_embeddings.layers.0.self_attn.linears.2.weight']))
    tiny = [k for k in names if float(np.linalg.norm(g['grad.' + k])) < 1e-3 * vnorm]
    assert 8 <= len(tiny) <= 9 and all(('linears.0.' in k or 'linears.1.' in k or 'q_lin' in k) for k in tiny), tiny
    assert all(float(named[k].grad.float().norm()) < 1e-3 * vnorm for k in tiny)
    bad = [(k, rel_l2(named[k].grad.float(), g['grad.' + k])) for k in names if k not in tiny]
    assert not [b for b in bad if b[1] > 4e-2], bad
    big = named['image_embeddings.image_embeddings.weight'].grad.float()
    assert rel_l2(big[:8], g['grad_rows8.image_embeddings.image_embeddings.weight']) < 4e-2
    assert abs(float(big.norm()) - float(g['grad_norm.image_embeddings.image_embeddings.weight'])) < 2e-2 * float(big.norm())
    # without the flag the refiner is not on the path (and a model without refiner layers refuses the flag)
    enc0 = m('crossfwd', stream_='img', x=x_img.cuda(), lengths=img_len.cuda(), langs=torch.ones((R, B), dtype=torch.long).cuda(),
             causal=False, image_loc=loc.cuda(), refine_image=False)
    assert rel_l2(enc0.float(), g['enc']) > 5e-2


@pytest.mark.gpu
def test_bart_mlm_step_vs_oracle():
    """Text infilling (xtrainer.py:1595-1646): the stream batch -> bart_token_mask_sent (bit-exact with the reference:
    tests/test_host_data.py) -> encoder over the <mask>ed sentence -> the whole sentence decoded with teacher forcing.  Loss and
    gradients against the oracle's autograd on the batch the step built (re-built here from the same seeds)."""
    import random
    from m3p_amd import masking
    from m3p_amd.model.transformer import TransformerModel
    from m3p_amd.trainer import XTrainer
    P, sd, x1_, len1_, _, _ = synth.mt_case()
    _step_params(P, use_noise=False, word_pred=0.15, sample_alpha=0, word_mask=0.8, word_keep=0.1, word_rand=0.1)
    x, lengths = x1_, len1_                    # (14, 6) sentences framed by EOS, as a stream / sentence batch holds them

    class Stream:
        def get_iterator(self, shuffle=True):
            return iter([(x, lengths)])

    torch.manual_seed(0)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True).cuda()
    m.load_state_dict(sd, strict=False)
    tr = XTrainer(m, {'mono_stream': {'zh': {'train': Stream()}}}, P)
    names = ['cross_lang_embeddings.weight', 'attentions.1.v_lin.weight', 'encoder_attn.0.q_lin.weight', 'encoder_attn.1.out_lin.weight',
             'ffns.0.lin2.weight', 'layer_norm15.1.bias', 'layer_norm_emb.weight', 'pred_layer.proj.bias']
    grads = _grads_at_step(m, tr.optimizers['model'], names)
    np.random.seed(91); random.seed(91); torch.manual_seed(91)
    loss = tr.bart_mlm_step('zh', None, 1.0)
    np.random.seed(91); random.seed(91); torch.manual_seed(91)
    x1, len1, x2, len2, y, pred_mask, _ = masking.bart_token_mask_sent(x, lengths, P)
    assert int((x1 == P.mask_index).sum()) == x.size(1)
    ref = {k: v.clone().requires_grad_(k in names) for k, v in sd.items()}
    enc = ref_cpu.crossfwd_text(ref, P.n_layers, P.n_heads, x1, len1, langs=x1.clone().fill_(1)).transpose(0, 1)
    dec = ref_cpu.decoder_crossfwd(ref, P.n_layers, P.n_heads, x2, len2, enc, len1, langs=x2.clone().fill_(1))
    o = ref_cpu.predict_mlm(ref, dec, pred_mask, y)
    o = o[1] if isinstance(o, tuple) else o
    o.backward()
    assert abs(float(loss) - float(o.detach())) < 5e-3
    bad = [(k, rel_l2(grads[k], ref[k].grad.numpy())) for k in names]
    assert not [b for b in bad if b[1] > 4e-2], bad
    assert 'M-BART-zh' in tr.stats and tr.stats['processed_w'] == int(pred_mask.sum()) == int((lengths - 1).sum())


def test_oracle_mass_step_matches_the_reference():
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'mass_step.npz'))
    P, sd, _, _, _, _ = synth.mt_case()
    t = lambda k: torch.from_numpy(g[k])      # noqa: E731
    x1, len1, x2, len2, y, pred_mask, pos = (t(k) for k in ('x1', 'len1', 'x2', 'len2', 'y', 'pred_mask', 'pos'))
    enc = ref_cpu.crossfwd_text(sd, P.n_layers, P.n_heads, x1, len1, langs=x1.clone().fill_(1)).transpose(0, 1)
    dec = ref_cpu.decoder_crossfwd(sd, P.n_layers, P.n_heads, x2, len2, enc, len1, positions=pos, langs=x2.clone().fill_(1),
                                   enc_mask=x1.ne(P.mask_index).t())
    assert np.abs(enc.numpy() - g['enc1']).max() < 2e-5 and np.abs(dec.numpy() - g['dec2']).max() < 2e-5
    loss = ref_cpu.predict_mlm(sd, dec, pred_mask, y)
    loss = loss[1] if isinstance(loss, tuple) else loss
    assert abs(float(loss) - float(g['loss'])) < 1e-5
    # the two inputs this step adds do matter: without them the decoder output is a different one
    plain = ref_cpu.decoder_crossfwd(sd, P.n_layers, P.n_heads, x2, len2, enc, len1, langs=x2.clone().fill_(1))
    nomask = ref_cpu.decoder_crossfwd(sd, P.n_layers, P.n_heads, x2, len2, enc, len1, positions=pos, langs=x2.clone().fill_(1))
    assert np.abs(plain.numpy() - g['dec2']).max() > 1e-2 and np.abs(nomask.numpy() - g['dec2']).max() > 1e-4


def test_drop_masked_source_packs_the_allowed_rows():
    from m3p_amd.model.transformer import TransformerModel
    src = torch.arange(2 * 5 * 3, dtype=torch.float32).view(2, 5, 3).requires_grad_(True)
    enc_mask = torch.tensor([[1, 0, 1, 1, 0], [0, 1, 1, 1, 1]], dtype=torch.bool)
    packed, n = TransformerModel._drop_masked_source(src, torch.tensor([4, 5]), enc_mask)
    assert n.tolist() == [3, 4]
    assert torch.equal(packed[0, :3], src[0, [0, 2, 3]].detach()) and float(packed[0, 3:].abs().sum()) == 0
    assert torch.equal(packed[1, :4], src[1, 1:].detach()) and float(packed[1, 4:].abs().sum()) == 0
    packed.sum().backward()
    assert src.grad[:, :, 0].tolist() == [[1, 0, 1, 1, 0], [0, 1, 1, 1, 1]]


@pytest.mark.gpu
def test_mass_step_vs_reference():
    """bart_mass_step's loss path: explicit decoder positions (the span's original positions) and enc_mask (the source's <mask>
    positions are not attended: the allowed source rows are packed to the front instead of a second mask in the kernels) -
    encoder / decoder outputs, loss and gradients against the reference's (tests/golden/mass_step.npz), incl. the position table,
    whose gradient now comes from two passes and the explicit positions."""
    from m3p_amd.model.transformer import TransformerModel
    from m3p_amd.trainer import XTrainer
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'mass_step.npz'))
    P, sd, _, _, _, _ = synth.mt_case()
    _step_params(P)
    t = lambda k: torch.from_numpy(g[k])      # noqa: E731
    x1, len1, x2, len2, y, pred_mask, pos = (t(k) for k in ('x1', 'len1', 'x2', 'len2', 'y', 'pred_mask', 'pos'))
    torch.manual_seed(0)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True).cuda()
    m.load_state_dict(sd, strict=False)
    m.train()
    enc = m('crossfwd', stream_='text', x=x1.cuda(), lengths=len1.cuda(), langs=x1.clone().fill_(1).cuda(), causal=False).transpose(0, 1)
    dec = m('crossfwd', stream_='text', x=x2.cuda(), lengths=len2.cuda(), langs=x2.clone().fill_(1).cuda(), causal=True, src_enc=enc,
            src_len=len1.cuda(), positions=pos.cuda(), enc_mask=x1.ne(P.mask_index).t().cuda())
    assert rel_l2(enc.float(), g['enc1']) < 1e-2 and rel_l2(dec.float(), g['dec2']) < 1e-2
    del enc, dec
    m.arena().zero_grad()
    tr = XTrainer(m, {}, P)
    names = [k[5:] for k in g.files if k.startswith('grad.')]
    grads = _grads_at_step(m, tr.optimizers['model'], names + ['embeddings.weight'])
    loss = tr.mass_step_on_batch(x1, len1, x2, len2, y, pred_mask, pos, 'zh', None, 1.0)
    assert abs(float(loss) - float(g['loss'])) < 5e-3
    enorm = float(grads.pop('embeddings.weight').norm())
    assert abs(enorm - float(g['grad_norm.embeddings.weight'])) < 4e-2 * enorm
    bad = [(k, rel_l2(v, g['grad.' + k])) for k, v in grads.items()]
    assert not [b for b in bad if b[1] > 4e-2], bad
    assert 'M-MASS-zh' in tr.stats and tr.stats['processed_w'] == int(pred_mask.sum())


@pytest.mark.gpu
def test_bart_img_step_vs_oracle():
    """Image denoising (xtrainer.py:1746-1808): caption_collate batch -> bart_img_noise (bit-exact with the reference:
    tests/test_host_data.py) -> the captioning pass on the shortened, partly blanked region set.  Loss and gradients against the
    oracle's autograd on the noised batch (re-built here from the same seeds)."""
    import random
    from m3p_amd import masking
    from m3p_amd.model.transformer import TransformerModel
    from m3p_amd.trainer import XTrainer
    P, sd, x_img, loc, _, x2, len2 = synth.ic_case()
    _step_params(P, is_generation=True, is_mt=False, is_pretrain=False, is_slide=False, n_gpu_per_node=1, num_workers=0, batch_size=6,
                 ft_lgs=[])
    R, B = x_img.shape[0], x_img.shape[1]
    feats = (x_img / x_img.norm(dim=-1, keepdim=True)).transpose(0, 1).contiguous()          # (B, R, 2048), unit rows

    class Items(torch.utils.data.Dataset):
        def __len__(self):
            return B

        def __getitem__(self, i):
            return (x2[1:int(len2[i]) - 1, i].numpy(), feats[i:i + 1], torch.ones(1, R, dtype=torch.long),
                    loc[:, i].unsqueeze(0).contiguous(), i)

    torch.manual_seed(0)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True).cuda()
    m.load_state_dict(sd, strict=False)
    tr = XTrainer(m, {'cross_modal': {('coco', 'img'): {'train': Items()}}}, P)
    names = ['image_embeddings.image_embeddings.bias', 'image_embeddings.LayerNorm.weight', 'attentions.0.v_lin.weight',
             'encoder_attn.1.k_lin.weight', 'ffns.1.lin1.weight', 'pred_layer.proj.bias']
    grads = _grads_at_step(m, tr.optimizers['model'], names)
    seen = {}
    real_get = tr.get_batch

    def get_batch(*a):
        b = real_get(*a)
        seen['ids'] = list(b[1][3])
        return b
    tr.get_batch = get_batch
    np.random.seed(17); random.seed(17)
    loss = tr.bart_img_step('coco', 'img', False, 1.0)
    order = seen['ids']
    np.random.seed(17); random.seed(17)
    f2, l2, m2 = masking.bart_img_noise(feats[order], loc.transpose(0, 1)[order], torch.ones(B, R, dtype=torch.long))
    n = f2.shape[1]
    assert 1 <= n < R and bool((f2.abs().sum(-1) == 0).any())           # fewer regions, some of them blank
    xs = x2[:, order].clone()
    xs[0] = 0                                                           # the collate frames sentences with BOS = 0
    ref = {k: v.clone().requires_grad_(k in names) for k, v in sd.items()}
    img_len = m2.sum(1)
    enc = ref_cpu.crossfwd_img(ref, P.n_layers, P.n_heads, f2.transpose(0, 1), img_len, l2.transpose(0, 1),
                               langs=torch.zeros((n, B), dtype=torch.long)).transpose(0, 1)
    dec = ref_cpu.decoder_crossfwd(ref, P.n_layers, P.n_heads, xs, len2[order], enc, img_len, langs=xs.clone().fill_(0))
    pred_mask, y = synth.mt_targets(xs, len2[order])
    o = ref_cpu.predict_mlm(ref, dec, pred_mask, y)
    o = o[1] if isinstance(o, tuple) else o
    o.backward()
    assert abs(float(loss) - float(o.detach())) < 5e-3
    bad = [(k, rel_l2(grads[k], ref[k].grad.numpy())) for k in names]
    assert not [b for b in bad if b[1] > 4e-2], bad
    assert 'IDA-coco' in tr.stats and tr.stats['processed_s'] == B


@pytest.mark.gpu
def test_free_lb_ic_step_tracks_the_reference_run():
    """free_lb_ic_step (xtrainer.py:2853-2962): three captioning passes with the caption's word rows (decoder text_embed) and the
    region features perturbed, each an optimizer step, an ascent step on both perturbations in between - against the same
    call on the reference's XTrainer on CPU under the same torch seed (tests/golden/freelb_ic_step.npz)."""
    from m3p_amd.model.transformer import TransformerModel
    from m3p_amd.trainer import XTrainer
    G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'freelb_ic_step.npz'))
    P, sd, x_img, loc, img_len, x2, len2 = synth.ic_case()
    _step_params(P, ft_lgs=[], free_text=True, free_img=True, cross_modal_steps=[('coco', 'img')], batch_size=x2.size(1))
    R, B = x_img.shape[0], x_img.shape[1]
    torch.manual_seed(0)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True).cuda()
    m.load_state_dict(sd, strict=False)
    tr = XTrainer(m, {}, P)
    batch = ((x2, len2), (x_img.transpose(0, 1).contiguous(), torch.ones(B, R, dtype=torch.long), loc.transpose(0, 1).contiguous(), list(range(B))))
    tr.get_batch = lambda *a, **k: batch
    own = dict(m.named_parameters())
    names = [k[6:] for k in G.files if k.startswith('dnorm/')]
    before = {k: own[k].detach().clone() for k in names}
    torch.manual_seed(777)
    loss = tr.free_lb_ic_step('coco', 'img', 1.0)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(G['loss'])) < 1e-2
    group = tr.optimizers['model'].param_groups[0]
    assert group['num_updates'] == int(G['n_updates']) == 3 and abs(group['lr'] - float(G['lr'])) < 1e-15
    assert [tr.stats['processed_s'], tr.stats['processed_w']] == G['processed'].tolist()
    for k in names:
        moved = float((own[k].detach() - before[k]).norm())
        assert abs(moved - float(G['dnorm/' + k])) < 0.1 * float(G['dnorm/' + k]), (k, moved, float(G['dnorm/' + k]))
    assert 'FRLB-IC-coco-img' in tr.stats


@pytest.mark.gpu
def test_decoder_text_embed_and_region_feature_gradients_vs_oracle():
    """What the FreeLB captioning step ascends along: d loss / d text_embed through DecoderFn and d loss / d x_img through the
    image stream, against the oracle's autograd (and text_embed = Emb[x] reproduces the plain pass)."""
    from m3p_amd.model.transformer import TransformerModel
    P, sd, x_img, loc, img_len, x2, len2 = synth.ic_case()
    _step_params(P)
    R, B = x_img.shape[0], x_img.shape[1]
    torch.manual_seed(0)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True).cuda()
    m.load_state_dict(sd, strict=False)
    m.train()
    P.dropout = P.attention_dropout = 0
    langs, langs_img = x2.clone().fill_(0), torch.zeros((R, B), dtype=torch.long)
    pred_mask, y = synth.mt_targets(x2, len2)

    def run(model_rows, feats):
        enc = m('crossfwd', stream_='img', x=feats, lengths=img_len.cuda(), langs=langs_img.cuda(), causal=False, image_loc=loc.cuda(),
                refine_image=False).transpose(0, 1)
        dec = m('crossfwd', stream_='text', x=x2.cuda(), lengths=len2.cuda(), langs=langs.cuda(), causal=True, src_enc=enc, src_len=img_len.cuda(),
                text_embed=model_rows)
        return m('predict', tensor=dec, pred_mask=pred_mask.cuda(), y=y.cuda(), get_scores=False)[1]
    m.arena().zero_grad()
    plain = run(None, x_img.cuda())
    rows = sd['embeddings.weight'][x2.t()].cuda().requires_grad_(True)
    feats = x_img.cuda().requires_grad_(True)
    loss = run(rows, feats)
    assert abs(float(loss) - float(plain)) < 2e-3
    loss.backward()
    ref_rows = sd['embeddings.weight'][x2.t()].clone().requires_grad_(True)
    ref_feats = x_img.clone().requires_grad_(True)
    enc = ref_cpu.crossfwd_img(sd, P.n_layers, P.n_heads, ref_feats, img_len, loc, langs=langs_img).transpose(0, 1)
    dec = ref_cpu.decoder_crossfwd(sd, P.n_layers, P.n_heads, x2, len2, enc, img_len, langs=langs, text_embed=ref_rows)
    o = ref_cpu.predict_mlm(sd, dec, pred_mask, y)
    o = o[1] if isinstance(o, tuple) else o
    o.backward()
    assert abs(float(loss) - float(o.detach())) < 5e-3
    assert rel_l2(rows.grad.float(), ref_rows.grad) < 5e-2 and rel_l2(feats.grad.float(), ref_feats.grad) < 5e-2
