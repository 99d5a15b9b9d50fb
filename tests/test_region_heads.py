"""Masked-region objectives: MRM (BertPredictionHeadTransform + ObjPredLayer, CE with
ignore_index) and MRFR (mrfr_dense + masked MSE) on the MI355X against the golden vectors recorded from
the reference (tests/golden/cfg1_region_heads.npz) and, end to end, against the oracle."""
import os

import numpy as np
import pytest
import torch

from m3p_amd import synth
from tests.util import rel_l2

pytestmark = pytest.mark.gpu


def _model(cfg):
    from m3p_amd.model.transformer import TransformerModel
    P = synth.model_params(cfg['emb_dim'], cfg['n_heads'], cfg['n_layers'], cfg['n_words'])
    torch.manual_seed(0)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True)
    sd = dict(synth.golden_state_dict(synth.hot_param_shapes(P)))
    rsd = synth.golden_state_dict(synth.region_head_param_shapes(P), seed=4321, pad_index=None)
    sd.update(rsd)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected
    return m.cuda(), P, sd, list(rsd.keys())


def test_region_heads_vs_reference_golden(golden_dir):
    """The heads alone, fed with the reference's own encoder output (bf16-rounded): losses and gradients."""
    from m3p_amd import functional as Fn
    g = dict(np.load(os.path.join(golden_dir, 'cfg1_region_heads.npz')))
    cfg = synth.CONFIGS['cfg1']
    m, P, sd, rnames = _model(cfg)
    tg = synth.make_region_targets(cfg['R'], cfg['B'])
    x = torch.from_numpy(g['img_out']).to(torch.bfloat16).cuda().requires_grad_(True)       # (B, R, d)
    m.arena().zero_grad()
    mrm = Fn.mrm_head(m, x, tg['obj_labels'].reshape(-1))
    mrfr = Fn.mrfr_head(m, x, tg['obj_labels'], tg['ori_att_feats'])
    assert abs(float(mrm) - float(g['mrm_loss'])) < 5e-3 and abs(float(mrfr) - float(g['mrfr_loss'])) < 5e-4
    (mrm + mrfr).backward()
    assert rel_l2(x.grad.float(), g['d_img_out']) < 5e-2
    own = dict(m.named_parameters())
    for k in rnames:
        assert rel_l2(own[k].grad, g['grad/' + k]) < 5e-2, k


def test_pretrain_step_with_mrm_and_mrfr_matches_oracle():
    """XTrainer.pretrain_under_step with the MLM + MRM + MRFR + ITM objective: losses and updated parameters
    against the oracle's train step on the same batch."""
    from oracle import ref_cpu as O
    from m3p_amd.trainer import XTrainer
    cfg = synth.CONFIGS['cfg1']
    m, P, sd, rnames = _model(cfg)
    batch = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'])
    tg = synth.make_region_targets(cfg['R'], cfg['B'])
    batch.update(tg)
    # oracle losses
    osd = {k: v.clone() for k, v in sd.items()}
    res = O.pretrain_losses(osd, cfg['n_layers'], cfg['n_heads'], batch, cfg['R'], with_mrm=True, with_mrfr=True)
    for k, v in dict(optimizer='adam_inverse_sqrt,beta1=0.9,beta2=0.98,lr=0.0001', clip_grad_norm=5, amp=-1, fp16=False,
                     accumulate_gradients=1, multi_gpu=False, epoch_size=100, cross_mlm_steps=[('google', 'img')],
                     cross_mrm_steps=[('google', 'img')], cross_mrfr_steps=[('google', 'img')], cross_clcm_steps=[], sample_n=2,
                     refine_image=False, multi_cls_loss_weight=0, bin_cls_loss_weight=1, batch_size=cfg['B'],
                     dump_path='/nonexistent_m3p_dump').items():
        setattr(P, k, v)
    trainer = XTrainer(m, {}, P)
    B, R = cfg['B'], cfg['R']
    img = batch['x_img'].transpose(0, 1).contiguous()                      # (B, R, 2048)
    loc = batch['image_loc'].transpose(0, 1).contiguous()
    tup = ((batch['x'], batch['lengths'], batch['x_labels']),
           (img, torch.ones(B, R, dtype=torch.long), loc, tg['obj_labels'], batch['pos_labels'].tolist(), tg['ori_att_feats'], None))
    trainer.pretrain_under_step(tup, 'google', 't2i', 'en', 1.0, 1.0, 1.0, 1.0)
    assert abs(float(trainer.stats['MRM-google'][-1]) - float(res['mrm'])) < 5e-3
    assert abs(float(trainer.stats['MRFR-google'][-1]) - float(res['mrfr'])) < 5e-4
    assert abs(float(trainer.stats['CMLM-google'][-1]) - float(res['mlm'])) < 5e-3
    # the region-head parameters moved (Adam touched them) and stayed finite
    own = dict(m.named_parameters())
    for k in rnames:
        assert torch.isfinite(own[k]).all()
        assert float((own[k].detach().cpu() - sd[k]).abs().max()) > 0, k


def test_clcm_second_pass_vs_reference_golden(golden_dir):
    """predict(is_clcm=True) on a second jointfwd (regions + the other caption): scores, BCE and gradients of the
    second head and of encoder parameters against the reference's own run (cfg1_clcm.npz)."""
    g = dict(np.load(os.path.join(golden_dir, 'cfg1_clcm.npz')))
    cfg = synth.CONFIGS['cfg1']
    from m3p_amd.model.transformer import TransformerModel
    P = synth.model_params(cfg['emb_dim'], cfg['n_heads'], cfg['n_layers'], cfg['n_words'])
    torch.manual_seed(0)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True)
    sd = dict(synth.golden_state_dict(synth.hot_param_shapes(P)))
    sd.update(synth.golden_state_dict(synth.clcm_head_param_shapes(P), seed=9753, pad_index=None))
    assert not m.load_state_dict(sd, strict=False)[1]
    m = m.cuda()
    batch = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'])
    b2 = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'], seed=8642)
    dev = 'cuda'
    m.eval()
    m.arena().zero_grad()
    out2 = m('jointfwd', x=b2['x'].to(dev), lengths=b2['lengths'].to(dev), x_img=batch['x_img'].to(dev),
             lengths_img=batch['lengths_img'].to(dev), causal=False, langs=None, image_loc=batch['image_loc'].to(dev),
             refine_image=False)
    rel2 = m('predict', tensor=out2.transpose(0, 1), is_clcm=True)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(rel2.view(-1).float(),
                                                                torch.from_numpy(g['clcm_labels']).float().to(dev))
    assert rel_l2(rel2.detach().float().cpu(), g['rel2']) < 2e-2 and abs(float(loss) - float(g['clcm_loss'])) < 5e-3
    loss.backward()
    own = dict(m.named_parameters())
    for k in [k[5:] for k in g if k.startswith('grad/')]:
        assert rel_l2(own[k].grad, g['grad/' + k]) < 5e-2, k


def test_i2t_pretrain_step_with_clcm_matches_oracle():
    """pretrain_under_step(task_name='i2t') with cross_clcm_steps: the CLCM loss logged by the trainer equals the
    oracle's second-pass BCE on the same batch (MLM + ITM still there)."""
    from oracle import ref_cpu as O
    from m3p_amd.model.transformer import TransformerModel
    from m3p_amd.trainer import XTrainer
    cfg = synth.CONFIGS['cfg1']
    P = synth.model_params(cfg['emb_dim'], cfg['n_heads'], cfg['n_layers'], cfg['n_words'])
    torch.manual_seed(0)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True)
    sd = dict(synth.golden_state_dict(synth.hot_param_shapes(P)))
    sd.update(synth.golden_state_dict(synth.clcm_head_param_shapes(P), seed=9753, pad_index=None))
    assert not m.load_state_dict(sd, strict=False)[1]
    m = m.cuda()
    batch = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'])
    b2 = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'], seed=8642)
    clcm_labels = torch.tensor([1, 0, 0, 1, 1, 0, 1, 0])
    ref, _ = O.clcm_loss({k: v.clone() for k, v in sd.items()}, cfg['n_layers'], cfg['n_heads'], batch, b2['x'], b2['lengths'],
                         clcm_labels)
    for k, v in dict(optimizer='adam_inverse_sqrt,beta1=0.9,beta2=0.98,lr=0.0001', clip_grad_norm=5, amp=-1, fp16=False,
                     accumulate_gradients=1, multi_gpu=False, epoch_size=100, cross_mlm_steps=[('google', 'img')],
                     cross_mrm_steps=[], cross_mrfr_steps=[], cross_clcm_steps=[('google', 'img')], sample_n=2,
                     refine_image=False, multi_cls_loss_weight=0, bin_cls_loss_weight=1, batch_size=cfg['B'],
                     dump_path='/nonexistent_m3p_dump').items():
        setattr(P, k, v)
    trainer = XTrainer(m, {}, P)
    B, R = cfg['B'], cfg['R']
    img = batch['x_img'].transpose(0, 1).contiguous()
    loc = batch['image_loc'].transpose(0, 1).contiguous()
    tup = ((batch['x'], batch['lengths'], batch['x_labels']), (b2['x'], b2['lengths']),
           (clcm_labels, img, torch.ones(B, R, dtype=torch.long), loc, torch.full((B, R), -1), batch['pos_labels'].tolist(), None, None))
    trainer.pretrain_under_step(tup, 'google', 'i2t', 'en', 1.0, 1.0, 1.0, 1.0)
    assert abs(float(trainer.stats['CLCM-google'][-1]) - float(ref)) < 5e-3
    own = dict(m.named_parameters())
    assert float((own['pooled_layer2.dense.weight'].detach().cpu() - sd['pooled_layer2.dense.weight']).abs().max()) > 0


def test_predict_is_mrfr_is_the_bare_regression(golden_dir):
    """predict(tensor, is_mrfr=True) (transformer.py:1202-1204): mrfr_dense on every row handed in, against the oracle."""
    from oracle import ref_cpu as O
    g = dict(np.load(os.path.join(golden_dir, 'cfg1_region_heads.npz')))
    cfg = synth.CONFIGS['cfg1']
    m, P, sd, _ = _model(cfg)
    x = torch.from_numpy(g['img_out']).to(torch.bfloat16)                       # (B, R, d)
    reg = m('predict', tensor=x.cuda(), is_mrfr=True)
    ref = O.predict_mrfr(sd, x.float())
    assert tuple(reg.shape) == tuple(ref.shape) == (cfg['B'], cfg['R'], 2048)
    assert rel_l2(reg.float(), ref) < 6e-3
