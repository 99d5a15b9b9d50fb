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
