"""TLM batches behind ``mlm_step(lang1, lang2, ...)`` (xtrainer.py:485-509, :734-770; utils.py:324-349) and the valid-set
matching accuracy of ``evaluate_t2i`` / ``evaluate_i2t`` (xevaluator.py:1309-1417), against goldens recorded by running the
reference (tests/golden/tlm_step.npz, eval_understanding.npz; oracle/gen_goldens.py::gen_tlm_goldens / gen_eval_goldens).
CPU: the host logic bit for bit and the oracle restatement; GPU: the HIP path at the §8c bars."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from m3p_amd import masking, synth
from tests.util import rel_l2


@pytest.fixture(scope='module')
def G(golden_dir):
    return dict(np.load(os.path.join(golden_dir, 'tlm_step.npz')))


def _case():
    P, sd, x1, len1, x2, len2 = synth.mt_case()
    for k, v in synth.trainer_params(batch_size=x1.shape[1], langs=['en', 'zh'], mlm_steps=[('en', 'zh')], clm_steps=[]).items():
        setattr(P, k, v)
    P.pred_probs = torch.FloatTensor([P.word_mask, P.word_keep, P.word_rand])
    P.mask_scores = None
    return P, sd, x1, len1, x2, len2


def test_tlm_batch_is_the_reference_bit_for_bit(G):
    """generate_batch('en', 'zh') -> round_batch -> mask_out under the golden's seeds: same x, lengths, positions, language
    ids, prediction mask and targets."""
    from m3p_amd.trainer import Trainer
    P, sd, x1, len1, x2, len2 = _case()
    fake = SimpleNamespace(params=P, get_cross_lingual_batch=lambda name, l1, l2=None, stream=False: ((x1, len1), (x2, len2)))
    np.random.seed(77); torch.manual_seed(77)
    x, lengths, positions, langs, (l1, l2) = Trainer.generate_batch(fake, 'en', 'zh', 'pred')
    assert torch.equal(l1, len1) and torch.equal(l2, len2)
    x, lengths, positions, langs, _ = masking.round_batch(x, lengths, positions, langs, P)
    x, y, pred_mask = masking.mask_out(x, lengths, P)
    for got, key in ((x, 'x'), (lengths, 'lengths'), (positions, 'positions'), (langs, 'langs'), (pred_mask, 'pred_mask'), (y, 'y')):
        assert np.array_equal(got.numpy(), G[key]), key
    # positions restart behind the first sentence; the second sentence carries the second language
    b = 1
    assert int(positions[int(len1[b]), b]) == 0 and int(langs[int(len1[b]), b]) == 1 and int(langs[int(len1[b]) - 1, b]) == 0


def test_oracle_text_stream_with_positions_matches_the_reference(G):
    from oracle import ref_cpu as O
    P, sd, *_ = _case()
    x, lengths = torch.from_numpy(G['x']), torch.from_numpy(G['lengths'])
    out = O.crossfwd_text(sd, P.n_layers, P.n_heads, x, lengths, langs=torch.from_numpy(G['langs']),
                          positions=torch.from_numpy(G['positions']))
    assert float((out - torch.from_numpy(G['out'])).abs().max()) < 1e-4


def _gpu_model(P, sd):
    from m3p_amd.model.transformer import TransformerModel
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True)
    m.load_state_dict(sd, strict=False)
    return m.cuda()


@pytest.mark.gpu
def test_tlm_stream_and_step_on_the_gpu(G):
    from m3p_amd.trainer import XTrainer
    P, sd, x1, len1, x2, len2 = _case()
    m = _gpu_model(P, sd).train()
    x, lengths = torch.from_numpy(G['x']).cuda(), torch.from_numpy(G['lengths']).cuda()
    positions, langs = torch.from_numpy(G['positions']).cuda(), torch.from_numpy(G['langs']).cuda()
    pred_mask, y = torch.from_numpy(G['pred_mask']).cuda(), torch.from_numpy(G['y']).cuda()
    out = m('crossfwd', stream_='text', x=x, lengths=lengths, positions=positions, langs=langs, causal=False)
    assert rel_l2(out.float(), G['out']) < 1e-2
    _, loss = m('predict', tensor=out, pred_mask=pred_mask, y=y, get_scores=False)
    assert abs(float(loss) - float(G['loss'])) < 5e-3
    loss.backward()
    torch.cuda.synchronize()
    ar = m.arena()
    for k in ('position_embeddings.weight', 'cross_lang_embeddings.weight', 'layer_norm_emb.weight', 'attentions.0.q_lin.weight',
              'ffns.1.lin2.weight', 'pred_layer.proj.bias'):
        assert rel_l2(ar.g(k), G['grad.' + k]) < 5e-2, (k, rel_l2(ar.g(k), G['grad.' + k]))
    gn = float(ar.g('embeddings.weight').norm())
    assert abs(gn - float(G['grad_norm.embeddings.weight'])) < 5e-2 * float(G['grad_norm.embeddings.weight'])
    # a plain-position run differs (the second sentence would sit at positions len1.. instead of 0..)
    with torch.no_grad():
        plain = m('crossfwd', stream_='text', x=x, lengths=lengths, positions=None, langs=langs, causal=False)
    assert rel_l2(plain.float(), G['out']) > 5e-2

    # the trainer's own step: the same batch under the golden's seeds, one clipped optimizer step
    m2 = _gpu_model(P, sd)

    class _Para:
        def get_iterator(self, shuffle=True, group_by_size=False, n_sentences=-1):
            return iter([((x1, len1), (x2, len2))])
    tr = XTrainer(m2, {'para': {('en', 'zh'): {'train': _Para()}}}, P)
    np.random.seed(77); torch.manual_seed(77)
    tr.mlm_step('en', 'zh', 1.0)
    assert abs(float(tr.stats['MLM-en-zh'][-1]) - float(G['step_loss'])) < 5e-3
    assert abs(tr.optimizers['model'].param_groups[0]['lr'] - float(G['step_lr'])) < 1e-12
    assert tr.stats['processed_s'] == int(G['step_processed'][0]) and tr.n_sentences == int(G['step_processed'][2])
    named = dict(m2.named_parameters())
    for k in ('embeddings.weight', 'position_embeddings.weight', 'cross_lang_embeddings.weight', 'attentions.0.q_lin.weight'):
        assert abs(float(named[k].norm()) - float(G['step_pnorm/' + k])) < 1e-3 * float(G['step_pnorm/' + k]), k


@pytest.mark.gpu
@pytest.mark.parametrize('tag,sample_n,pretrain', [('pre2', 2, True), ('fin4', 4, False)])
def test_evaluate_t2i_i2t_match_the_reference(golden_dir, tag, sample_n, pretrain):
    from m3p_amd import evaluation as E
    from m3p_amd.model.transformer import TransformerModel
    g = dict(np.load(os.path.join(golden_dir, 'eval_understanding.npz')))
    cfg = synth.CONFIGS['cfg1']
    P = synth.model_params(cfg['emb_dim'], cfg['n_heads'], cfg['n_layers'], cfg['n_words'])
    P.sample_n, P.is_pretrain, P.refine_image = sample_n, pretrain, False
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True)
    m.load_state_dict(synth.golden_state_dict(synth.hot_param_shapes(P)), strict=False)
    m = m.cuda().train()                      # (the evaluation switches to eval mode itself and back)
    B, R = cfg['B'], cfg['R']
    batch = synth.make_batch(cfg['T'], R, B, cfg['n_words'], cfg['n_pred'], seed=int(g[tag + '.seed']))
    img = batch['x_img'].transpose(0, 1).contiguous()
    loc = batch['image_loc'].transpose(0, 1).contiguous()
    mask = torch.ones(B, R, dtype=torch.long)
    pos = g[tag + '.pos'].tolist()
    if pretrain:
        t2i = ((batch['x'], batch['lengths'], batch['x_labels']), (img, mask, loc, torch.full((B, R), -1), pos, img.clone(), list(range(B))))
        i2t = ((batch['x'], batch['lengths'], batch['x_labels']), (batch['x'], batch['lengths']),
               (torch.zeros(B), img, mask, loc, torch.full((B, R), -1), pos, img.clone(), list(range(B))))
    else:
        t2i = i2t = ((batch['x'], batch['lengths'], torch.zeros_like(batch['x'])), (img, mask, loc, pos, list(range(B))))
    # the golden batch was chosen for its top-2 margins (>= 0.02): bf16 scores must land on the same argmax
    with torch.no_grad():
        m.eval()
        enc = m('jointfwd', x=batch['x'].cuda(), lengths=batch['lengths'].cuda(), x_img=batch['x_img'].cuda(),
                lengths_img=batch['lengths_img'].cuda(), causal=False, langs=None, image_loc=batch['image_loc'].cuda(), refine_image=False)
        sc = m('predict', tensor=enc.transpose(0, 1), is_relation=True).view(-1, sample_n).float().cpu()
        m.train()
    assert float((sc - torch.from_numpy(g[tag + '.scores'])).abs().max()) < 0.4 * float(g[tag + '.margin'].min())
    assert E.evaluate_t2i(m, P, t2i) == tuple(int(v) for v in g[tag + '.t2i'])
    assert E.evaluate_i2t(m, P, i2t) == tuple(int(v) for v in g[tag + '.i2t'])
    assert m.training
    scores = E.evaluate_understanding_tasks(m, P, [(t2i, i2t)], {}, 'valid', 'coco', 'img')
    assert scores['valid_coco-img_rel_t2i_acc'] == 100.0 * g[tag + '.t2i'][0] / g[tag + '.t2i'][1]
