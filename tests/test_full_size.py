"""Size-independent properties at BASELINE.json's full cfg2 size (12L/768d/12h, 36 regions + 128 tokens,
V = 250 002, batch 256 on one MI355X), where the oracle cannot be run in seconds:

* the batch-mean losses and every gradient of a 256-sequence step equal the average of the two 128-sequence
  half-batch steps (the identity data parallelism rests on: equal MLM target counts per half);
* rows behind a sequence's length are exactly zero in the encoder output (transformer.py:958);
* the initial MLM loss of a N(0, d^-1/2)-initialised tied embedding sits at ln(V) + small.
"""
import math

import pytest
import torch

from m3p_amd import synth
from tests.util import rel_l2

pytestmark = pytest.mark.gpu


def _losses(m, batch, R, sl, sample_n=2):
    dev = 'cuda'
    x, lengths = batch['x'][:, sl].to(dev), batch['lengths'][sl].to(dev)
    out = m('jointfwd', x=x, lengths=lengths, x_img=batch['x_img'][:, sl].to(dev), lengths_img=batch['lengths_img'][sl].to(dev),
            causal=False, langs=None, image_loc=batch['image_loc'][:, sl].to(dev), refine_image=False)
    pm = batch['pred_mask'][:, sl].to(dev)
    y = batch['x_labels'][:, sl][batch['pred_mask'][:, sl]].to(dev)
    _, mlm = m('predict', tensor=out[R:], pred_mask=pm, y=y, get_scores=False)
    rel = m('predict', tensor=out.transpose(0, 1), is_relation=True)
    pos = batch['pos_labels'][sl.start // sample_n:sl.stop // sample_n].to(dev)
    onehot = torch.eye(sample_n, device=dev)[pos].reshape(-1)
    bce = torch.nn.functional.binary_cross_entropy_with_logits(rel.view(-1).float(), onehot)
    return out, mlm, bce


@pytest.mark.parametrize('B', [256, 1024])
def test_cfg2_full_size_batch_split_identity(B):
    """B = 256: BASELINE configs[1].  B = 1024: the per-GPU share of configs[2] (global batch 8192 over 8 MI355X)."""
    from m3p_amd.model.transformer import TransformerModel
    cfg = dict(synth.CONFIGS['cfg2'])
    cfg['B'] = B
    P = synth.model_params(cfg['emb_dim'], cfg['n_heads'], cfg['n_layers'], cfg['n_words'], dropout=0.0, attention_dropout=0.0)
    torch.manual_seed(1234)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True).cuda()
    m.train()
    B, R, T = cfg['B'], cfg['R'], cfg['T']
    # two ragged half-batches (each holds one full-length row, so either half alone is a valid batch) side by side
    ha = synth.make_batch(T, R, B // 2, cfg['n_words'], cfg['n_pred'], seed=77, ragged=True)
    hb = synth.make_batch(T, R, B // 2, cfg['n_words'], cfg['n_pred'], seed=78, ragged=True)
    batch = {k: torch.cat([ha[k], hb[k]], dim=1) for k in ('x', 'x_labels', 'pred_mask', 'x_img', 'image_loc')}
    batch.update({k: torch.cat([ha[k], hb[k]]) for k in ('lengths', 'lengths_img', 'pos_labels')})
    names = ['embeddings.weight', 'position_embeddings.weight', 'attentions.0.q_lin.weight', 'attentions.11.out_lin.weight',
             'ffns.5.lin1.weight', 'ffns.5.lin1.bias', 'ffns.11.lin2.weight', 'layer_norm2.7.weight', 'layer_norm1.0.bias',
             'image_embeddings.image_embeddings.weight', 'pooled_layer.dense.weight', 'pred_layer.proj.bias']
    own = dict(m.named_parameters())

    def run(sl):
        m.arena().zero_grad()
        out, mlm, bce = _losses(m, batch, R, sl)
        (mlm + bce).backward()
        torch.cuda.synchronize()
        return out, float(mlm.detach()), float(bce.detach()), {n: own[n].grad.detach().float().clone() for n in names}

    out, mlm, bce, g = run(slice(0, B))
    # padded positions are exactly zero
    tot = batch['lengths'] + R
    worst = max(float(out[int(tot[b]):, b].abs().max()) for b in range(B) if int(tot[b]) < out.shape[0])
    assert worst == 0.0
    # random init: the loss of near-uniform guessing, ln V + (logit variance) / 2
    assert 0.0 < mlm - math.log(cfg["n_words"]) < 3.0, mlm
    assert 0.3 < bce < 1.5, bce
    _, mlm_a, bce_a, ga = run(slice(0, B // 2))
    _, mlm_b, bce_b, gb = run(slice(B // 2, B))
    assert abs(0.5 * (mlm_a + mlm_b) - mlm) < 2e-3 and abs(0.5 * (bce_a + bce_b) - bce) < 2e-3
    bad = [(n, rel_l2(0.5 * (ga[n] + gb[n]), g[n])) for n in names]
    bad = [(n, e) for n, e in bad if e > 2e-2]
    assert not bad, bad


def test_translation_step_full_size_batch_split_identity():
    """The translation step (encoder pass + teacher-forced causal pass over it) at M3P-base size - 12 layers / 768 wide /
    12 heads (head dim 64), V = 250 002, 64 sentence pairs of up to 48 source / 32 target words: the loss of near-uniform
    guessing, finite gradients on every encoder-attention tensor, and loss / gradients of the whole batch equal to the
    average of its two halves (equal target-word counts per half)."""
    from m3p_amd.model.transformer import TransformerModel
    P = synth.model_params(768, 12, 12, 250002, n_langs=2, id2lang={0: 'en', 1: 'zh'}, lang2id={'en': 0, 'zh': 1},
                           mt_steps=[('en', 'zh')])
    torch.manual_seed(4321)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True).cuda()
    m.train()
    g = torch.Generator().manual_seed(5)
    B, T1, T2 = 64, 48, 32

    def sentences(T, lens):
        x = torch.randint(3, P.n_words, (T, B), generator=g)
        x[0] = synth.EOS
        for b in range(B):
            x[int(lens[b]) - 1, b] = synth.EOS
            x[int(lens[b]):, b] = synth.PAD
        return x
    len1 = torch.randint(T1 // 2, T1 + 1, (B,), generator=g)
    len1[0] = len1[B // 2] = T1
    len2 = torch.randint(T2 // 2, T2 + 1, (B // 2,), generator=g)
    len2[0] = T2
    len2 = torch.cat([len2, len2])                       # the same number of target words in both halves
    x1, x2 = sentences(T1, len1), sentences(T2, len2)
    names = ['encoder_attn.0.q_lin.weight', 'encoder_attn.11.k_lin.weight', 'encoder_attn.5.v_lin.bias', 'encoder_attn.7.out_lin.weight',
             'layer_norm15.3.weight', 'attentions.11.q_lin.weight', 'ffns.0.lin2.weight', 'cross_lang_embeddings.weight',
             'position_embeddings.weight', 'pred_layer.proj.bias']
    own = dict(m.named_parameters())

    def run(sl):
        m.arena().zero_grad()
        a, la, b_, lb = x1[:, sl].cuda(), len1[sl].cuda(), x2[:, sl].cuda(), len2[sl].cuda()
        Tb = int(lb.max())
        b_ = b_[:Tb]
        pred_mask, y = synth.mt_targets(x2[:Tb, sl], len2[sl])
        enc = m('crossfwd', stream_='text', x=a, lengths=la, langs=torch.zeros_like(a), causal=False).transpose(0, 1)
        dec = m('crossfwd', stream_='text', x=b_, lengths=lb, langs=torch.ones_like(b_), causal=True, src_enc=enc, src_len=la)
        _, loss = m('predict', tensor=dec, pred_mask=pred_mask.cuda(), y=y.cuda(), get_scores=False)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), {n: own[n].grad.detach().float().clone() for n in names}

    loss, gr = run(slice(0, B))
    assert 0.0 < loss - math.log(P.n_words) < 3.0, loss
    assert all(torch.isfinite(v).all() and float(v.abs().max()) > 0 for v in gr.values())
    la_, ga = run(slice(0, B // 2))
    lb_, gb = run(slice(B // 2, B))
    assert abs(0.5 * (la_ + lb_) - loss) < 2e-3
    bad = [(n, rel_l2(0.5 * (ga[n] + gb[n]), gr[n])) for n in names]
    bad = [(n, e) for n, e in bad if e > 3e-2]
    assert not bad, bad
