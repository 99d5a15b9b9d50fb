#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE itself (development container only).

    python3 -B oracle/gen_goldens.py            # writes tests/golden/*.npz

Imports microsoft/M3P from /root/reference (read-only mount, hence ``-B``), loads the
deterministic golden weights / synthetic batch from ``m3p_amd.synth`` and records what the
reference computes on the hot path:

  cfg1_model.npz   TransformerModel('jointfwd') output, MLM + ITM losses, relation scores,
                   per-parameter gradient norms (+ a few full gradients), parameters
                   after 1 and 3 AdamInverseSqrtWithWarmup steps with clip 5, lr sequence.
  cfg1_trainer.npz XTrainer.pretrain_under_step run end-to-end on CPU through three
                   container-only shims (stub ``apex``, ``Tensor.cuda`` = identity, a
                   fabricated batch tuple): logged losses, lr, parameter norms.
  units.npz        MultiHeadAttention / TransformerFFN / BertImageEmbeddings / gelu /
                   get_masks / Adam on small random inputs (per-kernel oracle pins).

The reference never travels to the GPU box: only these .npz fixtures (data, not source)
are committed.  /root/reference is not needed to *run* the tests.
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/M3P'
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
warnings.filterwarnings('ignore')

from m3p_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def build_reference_model(cfg, dropout=0.0):
    from src.model.transformer import TransformerModel
    P = synth.model_params(cfg['emb_dim'], cfg['n_heads'], cfg['n_layers'], cfg['n_words'],
                           dropout=dropout, attention_dropout=dropout)
    torch.manual_seed(0)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True)
    shapes = synth.hot_param_shapes(P)
    gsd = synth.golden_state_dict(shapes)
    own = dict(m.named_parameters())
    with torch.no_grad():
        for k, v in gsd.items():
            assert tuple(own[k].shape) == tuple(v.shape), (k, own[k].shape, v.shape)
            own[k].copy_(v)
    return m, P, list(shapes.keys())


def ref_losses(m, batch, R, sample_n=2):
    out = m('jointfwd', x=batch['x'], lengths=batch['lengths'], x_img=batch['x_img'],
            lengths_img=batch['lengths_img'], causal=False, langs=None,
            image_loc=batch['image_loc'], refine_image=False)
    scores, mlm = m('predict', tensor=out[R:], pred_mask=batch['pred_mask'], y=batch['y'], get_scores=True)
    rel = m('predict', tensor=out.transpose(0, 1), is_relation=True)
    onehot = torch.eye(sample_n)[batch['pos_labels']].reshape(-1)
    bce = torch.nn.functional.binary_cross_entropy_with_logits(rel.view(-1), onehot)
    ce = torch.nn.functional.cross_entropy(rel.view(-1, sample_n), batch['pos_labels'])
    return out, scores, mlm, rel, bce, ce


def gen_model_goldens():
    from src.optim import get_optimizer
    from torch.nn.utils import clip_grad_norm_
    cfg = synth.CONFIGS['cfg1']
    m, P, hot = build_reference_model(cfg)
    m.train()
    batch = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'])
    R = cfg['R']
    g = {}
    out, scores, mlm, rel, bce, ce = ref_losses(m, batch, R)
    g['out'] = out.detach().numpy()
    g['mlm_loss'] = mlm.detach().numpy()
    g['mlm_scores_rows8'] = scores[:8].detach().numpy()
    g['mlm_scores_sum'] = scores.double().sum().detach().numpy()
    g['rel_scores'] = rel.detach().numpy()
    g['itm_bce'] = bce.detach().numpy()
    g['itm_ce'] = ce.detach().numpy()

    # gradients of total = MLM + BCE (bin weight 1, multi weight 0: README default)
    named = dict(m.named_parameters())
    params = [named[k] for k in hot]
    opt = get_optimizer(params, 'adam_inverse_sqrt,beta1=0.9,beta2=0.98,lr=0.0001')
    lrs = [opt.param_groups[0]['lr']]
    full_grad_keys = ['layer_norm_emb.weight', 'layer_norm1.0.weight', 'layer_norm2.1.bias',
                      'attentions.0.q_lin.weight', 'attentions.0.k_lin.bias', 'attentions.1.v_lin.bias',
                      'ffns.0.lin1.bias', 'ffns.1.lin2.weight', 'image_embeddings.image_location_embeddings.weight',
                      'pooled_layer.dense.weight', 'seq_relationship.weight', 'pred_layer.proj.bias',
                      'position_embeddings.weight']
    for step in range(3):
        opt.zero_grad()
        out, scores, mlm, rel, bce, ce = ref_losses(m, batch, R)
        total = mlm + bce
        total.backward()
        if step == 0:
            for k in hot:
                g['gradnorm/' + k] = named[k].grad.norm().numpy()
            for k in full_grad_keys:
                g['grad/' + k] = named[k].grad.numpy().copy()
            rows = torch.unique(batch['x'].reshape(-1))[:16]
            g['grad_emb_rows_idx'] = rows.numpy()
            g['grad/embeddings.weight[rows]'] = named['embeddings.weight'].grad[rows].numpy().copy()
        g['total_loss_step%d' % step] = total.detach().numpy()
        gn = clip_grad_norm_(params, 5.0)
        g['gradnorm_total_step%d' % step] = np.asarray(float(gn))
        opt.step()
        lrs.append(opt.param_groups[0]['lr'])
        if step in (0, 2):
            for k in hot:
                g['param_norm_after%d/%s' % (step + 1, k)] = named[k].detach().norm().numpy()
            for k in ['layer_norm_emb.weight', 'attentions.0.q_lin.weight', 'ffns.1.lin2.bias']:
                g['param_after%d/%s' % (step + 1, k)] = named[k].detach().numpy().copy()
    g['lrs'] = np.asarray(lrs, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'cfg1_model.npz'), **g)
    print('cfg1_model.npz: mlm %.6f bce %.6f ce %.6f' % (float(g['mlm_loss']), float(g['itm_bce']), float(g['itm_ce'])))
    print('lrs', lrs)


def gen_trainer_goldens():
    """XTrainer.pretrain_under_step through the three container-only shims."""
    for name in ('apex', 'apex.amp', 'apex.parallel'):
        sys.modules.setdefault(name, types.ModuleType(name))
    torch.Tensor.cuda = lambda self, *a, **k: self
    import src.xtrainer as xt

    cfg = synth.CONFIGS['cfg1']
    m, P, hot = build_reference_model(cfg)
    extra = dict(
        langs=['en'], encoder_only=True, epoch_size=100, stopping_criterion='', amp=-1, fp16=False,
        accumulate_gradients=1, multi_gpu=False, local_rank=0, word_mask=0.8, word_keep=0.1, word_rand=0.1,
        validation_metrics='', dump_path='/nonexistent_m3p_dump', reload_checkpoint='',
        optimizer='adam_inverse_sqrt,beta1=0.9,beta2=0.98,lr=0.0001', use_memory=0, clip_grad_norm=5,
        pc_steps=[], ae_steps=[], mt_steps=[], mass_steps=[], bt_steps=[], cross_modal_steps=[],
        cross_rel_steps=[('google', 'img')], cross_mass_steps=[], cross_ae_steps=[], cross_gan_steps=[],
        cross_mlm_steps=[('google', 'img')], cross_mrm_steps=[], cross_mrfr_steps=[], cross_clcm_steps=[],
        max_region_num=cfg['R'], sample_n=2, is_latent=False, refine_image=False,
        multi_cls_loss_weight=0, bin_cls_loss_weight=1, batch_size=cfg['B'],
    )
    for k, v in extra.items():
        setattr(P, k, v)
    for lam in ('lambda_clm', 'lambda_mlm', 'lambda_pc', 'lambda_ae', 'lambda_mt', 'lambda_bt', 'lambda_mass',
                'lambda_ic', 'lambda_imlm', 'lambda_ida', 'lambda_tifg', 'lambda_rel', 'lambda_mrm',
                'lambda_mrfr', 'lambda_t2i', 'lambda_i2t'):
        setattr(P, lam, '1')
    tr = xt.XTrainer(m, {}, P)
    batch = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'])
    B, R = cfg['B'], cfg['R']
    img = batch['x_img'].transpose(0, 1).contiguous()        # (n, R, 2048)
    loc = batch['image_loc'].transpose(0, 1).contiguous()    # (n, R, 5)
    img_mask = torch.ones(B, R, dtype=torch.long)
    obj_labels = torch.full((B, R), -1, dtype=torch.long)
    pos_labels = batch['pos_labels'].tolist()
    tup = ((batch['x'], batch['lengths'], batch['x_labels']),
           (img, img_mask, loc, obj_labels, pos_labels, img.clone(), list(range(B))))
    g = {}
    named = dict(m.named_parameters())
    for step in range(2):
        tr.pretrain_under_step(tup, 'google', 't2i', 'en', 1.0, 1.0, 1.0, 1.0)
        g['cmlm_step%d' % step] = np.asarray(tr.stats['CMLM-google'][-1])
        g['t2i_step%d' % step] = np.asarray(tr.stats['t2i-google'][-1])
        g['lr_after%d' % step] = np.asarray(tr.optimizers['model'].param_groups[0]['lr'])
        for k in hot:
            g['param_norm_after%d/%s' % (step + 1, k)] = named[k].detach().norm().numpy()
    g['processed_s'] = np.asarray(tr.stats['processed_s'])
    g['processed_w'] = np.asarray(tr.stats['processed_w'])
    g['n_sentences'] = np.asarray(tr.n_sentences)
    np.savez_compressed(os.path.join(OUT, 'cfg1_trainer.npz'), **g)
    print('cfg1_trainer.npz:', {k: float(v) for k, v in g.items() if v.ndim == 0 and '/' not in k})


def gen_unit_goldens():
    import src.model.transformer as T
    from src.optim import Adam
    rs = np.random.RandomState(99)
    g = {}
    # gelu
    xs = torch.from_numpy(rs.standard_normal(257).astype(np.float32) * 3)
    g['gelu_x'] = xs.numpy(); g['gelu_y'] = T.gelu(xs).numpy()
    # get_masks
    lens = torch.tensor([5, 9, 1, 7])
    mask, am = T.get_masks(9, lens, False)
    g['masks_len'] = lens.numpy(); g['masks_mask'] = mask.numpy()
    # MultiHeadAttention, d=64, 2 heads, S=11, ragged key mask
    torch.manual_seed(1)
    mha = T.MultiHeadAttention(2, 64, dropout=0.0).eval()
    x = torch.from_numpy(rs.standard_normal((3, 11, 64)).astype(np.float32))
    km = torch.arange(11)[None, :] < torch.tensor([11, 6, 9])[:, None]
    g['mha_x'] = x.numpy(); g['mha_mask'] = km.numpy()
    for k, v in mha.state_dict().items():
        g['mha_sd/' + k] = v.numpy()
    g['mha_y'] = mha(x, km).detach().numpy()
    # TransformerFFN
    ffn = T.TransformerFFN(64, 256, 64, dropout=0.0, gelu_activation=True).eval()
    for k, v in ffn.state_dict().items():
        g['ffn_sd/' + k] = v.numpy()
    g['ffn_y'] = ffn(x).detach().numpy()
    # BertImageEmbeddings
    ie = T.BertImageEmbeddings(64, 2, 0.0).eval()
    feats = torch.from_numpy(rs.standard_normal((3, 5, 2048)).astype(np.float32))
    loc = torch.from_numpy(rs.uniform(size=(3, 5, 5)).astype(np.float32))
    for k, v in ie.state_dict().items():
        g['ie_sd/' + k] = v.numpy()
    g['ie_feats'] = feats.numpy(); g['ie_loc'] = loc.numpy()
    g['ie_y'] = ie(feats, loc).detach().numpy()
    # Adam (plain) with weight decay, 3 steps
    p = torch.nn.Parameter(torch.from_numpy(rs.standard_normal(37).astype(np.float32)))
    g['adam_p0'] = p.detach().numpy().copy()
    opt = Adam([p], lr=1e-2, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)
    grads = rs.standard_normal((3, 37)).astype(np.float32)
    g['adam_grads'] = grads
    for i in range(3):
        p.grad = torch.from_numpy(grads[i].copy())
        opt.step()
        g['adam_p%d' % (i + 1)] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'units.npz'), **g)
    print('units.npz ok')


def gen_text_and_itm_goldens():
    """crossfwd text stream (mlm_step) and the sample_n = 4 relation loss (t2i/i2t fine-tune)."""
    cfg = synth.CONFIGS['cfg1']
    m, P, hot = build_reference_model(cfg)
    m.eval()
    batch = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'])
    g = {}
    with torch.no_grad():
        out = m('crossfwd', stream_='text', x=batch['x'], lengths=batch['lengths'], positions=None, langs=None, causal=False)
        scores, mlm = m('predict', tensor=out, pred_mask=batch['pred_mask'], y=batch['y'], get_scores=True)
        g['text_out'] = out.numpy()
        g['text_mlm_loss'] = mlm.numpy()
        joint = m('jointfwd', x=batch['x'], lengths=batch['lengths'], x_img=batch['x_img'], lengths_img=batch['lengths_img'],
                  causal=False, langs=None, image_loc=batch['image_loc'], refine_image=False)
        rel = m('predict', tensor=joint.transpose(0, 1), is_relation=True)
        pos = torch.tensor([2, 0])      # B = 8 -> two groups of sample_n = 4
        g['rel4_pos'] = pos.numpy()
        g['rel4_ce'] = torch.nn.functional.cross_entropy(rel.view(-1, 4), pos).numpy()
        g['rel4_bce'] = torch.nn.functional.binary_cross_entropy_with_logits(rel.view(-1), torch.eye(4)[pos].reshape(-1)).numpy()
    np.savez_compressed(os.path.join(OUT, 'cfg1_text_itm.npz'), **g)
    print('cfg1_text_itm.npz: text mlm %.6f rel4 ce %.6f bce %.6f' % (float(g['text_mlm_loss']), float(g['rel4_ce']), float(g['rel4_bce'])))


def gen_region_head_goldens():
    """MRM + MRFR heads on the cfg1 batch: the reference's predict(is_obj=True) /
    predict(is_mrfr=True) and the trainer's masked MSE, with gradients of the head parameters and
    of the encoder output they read."""
    cfg = synth.CONFIGS['cfg1']
    m, P, hot = build_reference_model(cfg)
    m.eval()
    rshapes = synth.region_head_param_shapes(P)
    rsd = synth.golden_state_dict(rshapes, seed=4321, pad_index=None)
    own = dict(m.named_parameters())
    with torch.no_grad():
        for k, v in rsd.items():
            assert tuple(own[k].shape) == tuple(v.shape), (k, own[k].shape, v.shape)
            own[k].copy_(v)
    batch = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'])
    tg = synth.make_region_targets(cfg['R'], cfg['B'])
    R = cfg['R']
    for p_ in m.parameters():
        p_.grad = None
    out = m('jointfwd', x=batch['x'], lengths=batch['lengths'], x_img=batch['x_img'], lengths_img=batch['lengths_img'],
            causal=False, langs=None, image_loc=batch['image_loc'], refine_image=False)
    img_out = out[:R].transpose(0, 1).detach().clone().requires_grad_(True)
    scores, mrm = m('predict', tensor=img_out, pred_mask=None, y=tg['obj_labels'].view(-1), get_scores=False, is_obj=True)
    reg = m('predict', tensor=img_out, is_mrfr=True)
    mask = tg['obj_labels'].reshape(-1) != -1
    mrfr = torch.nn.functional.mse_loss(reg.reshape(-1, 2048)[mask], tg['ori_att_feats'].reshape(-1, 2048)[mask])
    (mrm + mrfr).backward()
    g = {'img_out': img_out.detach().numpy(), 'mrm_scores': scores.detach().numpy(), 'mrm_loss': mrm.detach().numpy(),
         'mrfr_reg': reg.detach().numpy(), 'mrfr_loss': mrfr.detach().numpy(), 'd_img_out': img_out.grad.numpy()}
    for k in rshapes:
        g['grad/' + k] = own[k].grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'cfg1_region_heads.npz'), **g)
    print('cfg1_region_heads.npz: mrm %.6f mrfr %.6f' % (float(mrm), float(mrfr)))


def gen_clcm_goldens():
    """CLCM second pass: reference jointfwd on (regions, second caption) + predict(is_clcm=True)
    + BCE, with the gradients of the second head and a few encoder parameters."""
    cfg = synth.CONFIGS['cfg1']
    m, P, hot = build_reference_model(cfg)
    m.eval()
    cshapes = synth.clcm_head_param_shapes(P)
    csd = synth.golden_state_dict(cshapes, seed=9753, pad_index=None)
    own = dict(m.named_parameters())
    with torch.no_grad():
        for k, v in csd.items():
            assert tuple(own[k].shape) == tuple(v.shape), (k, own[k].shape, v.shape)
            own[k].copy_(v)
    batch = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'])
    b2 = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'], seed=8642)   # the other caption
    clcm_labels = torch.tensor([1, 0, 0, 1, 1, 0, 1, 0])
    for p_ in m.parameters():
        p_.grad = None
    out2 = m('jointfwd', x=b2['x'], lengths=b2['lengths'], x_img=batch['x_img'], lengths_img=batch['lengths_img'],
             causal=False, langs=None, image_loc=batch['image_loc'], refine_image=False)
    rel2 = m('predict', tensor=out2.transpose(0, 1), is_clcm=True)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(rel2.view(-1), clcm_labels.float())
    loss.backward()
    g = {'clcm_labels': clcm_labels.numpy(), 'rel2': rel2.detach().numpy(), 'clcm_loss': loss.detach().numpy()}
    for k in cshapes:
        g['grad/' + k] = own[k].grad.numpy()
    for k in ('layer_norm2.1.weight', 'attentions.0.q_lin.bias', 'ffns.1.lin2.bias'):
        g['grad/' + k] = own[k].grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'cfg1_clcm.npz'), **g)
    print('cfg1_clcm.npz: clcm bce %.6f' % float(loss))


def gen_refiner_goldens():
    """AoA refiner: the reference's jointfwd(refine_image=True) with params.refine_layers = 2 on the
    cfg1 batch (eval mode: the refiner's dropouts are hard-wired to 0.1 in train mode), MLM + ITM losses and the
    gradients of every refiner parameter and of a few parameters up- and downstream of it; plus the refiner
    module alone on a random (B, R, d) input with a ragged region mask."""
    from src.model.transformer import TransformerModel
    cfg = synth.CONFIGS['cfg1']
    P = synth.model_params(cfg['emb_dim'], cfg['n_heads'], cfg['n_layers'], cfg['n_words'], refine_layers=2)
    torch.manual_seed(0)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True)
    shapes = synth.hot_param_shapes(P)
    rshapes = synth.refiner_param_shapes(P)
    own = dict(m.named_parameters())
    with torch.no_grad():
        for sd_ in (synth.golden_state_dict(shapes), synth.golden_state_dict(rshapes, seed=2468, pad_index=None)):
            for k, v in sd_.items():
                assert tuple(own[k].shape) == tuple(v.shape), (k, own[k].shape, v.shape)
                own[k].copy_(v)
    assert set(n for n in own if n.startswith('refine_embeddings.')) == set(rshapes), 'refiner parameter enumeration'
    m.eval()
    batch = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'])
    R = cfg['R']
    out = m('jointfwd', x=batch['x'], lengths=batch['lengths'], x_img=batch['x_img'], lengths_img=batch['lengths_img'],
            causal=False, langs=None, image_loc=batch['image_loc'], refine_image=True)
    _, mlm = m('predict', tensor=out[R:], pred_mask=batch['pred_mask'], y=batch['y'], get_scores=False)
    rel = m('predict', tensor=out.transpose(0, 1), is_relation=True)
    onehot = torch.eye(2)[batch['pos_labels']].reshape(-1)
    bce = torch.nn.functional.binary_cross_entropy_with_logits(rel.view(-1), onehot)
    (mlm + bce).backward()
    g = {'out': out.detach().numpy(), 'mlm_loss': mlm.detach().numpy(), 'itm_bce': bce.detach().numpy()}
    for k in list(rshapes) + ['image_embeddings.image_embeddings.weight', 'image_embeddings.LayerNorm.weight',
                              'image_embeddings.image_location_embeddings.weight', 'layer_norm_emb.weight',
                              'attentions.0.q_lin.weight', 'position_embeddings.weight']:
        # (copies: the module-alone backward below accumulates into the same .grad storage)
        g['grad/' + k] = (own[k].grad if k != 'position_embeddings.weight' else own[k].grad[:R + cfg['T']]).clone().numpy()
    # the module alone, ragged mask
    rs = np.random.RandomState(97)
    B = cfg['B']
    xin = torch.from_numpy(rs.standard_normal((B, R, cfg['emb_dim'])).astype(np.float32)).requires_grad_(True)
    lens = torch.from_numpy(rs.randint(R // 2, R + 1, size=B)).long()
    mask = torch.arange(R)[None, :] < lens[:, None]
    y = m.refine_embeddings(xin, mask)
    wgt = torch.from_numpy(rs.standard_normal(tuple(y.shape)).astype(np.float32))
    (y * wgt).sum().backward()
    g.update({'unit_x': xin.detach().numpy(), 'unit_lens': lens.numpy(), 'unit_y': y.detach().numpy(), 'unit_w': wgt.numpy(),
              'unit_dx': xin.grad.numpy()})
    np.savez_compressed(os.path.join(OUT, 'cfg1_refiner.npz'), **g)
    print('cfg1_refiner.npz: mlm %.6f itm %.6f' % (float(mlm), float(bce)))


def gen_state_dict_enumeration():
    """Checkpoint interop: every key and shape of the reference model's state_dict() - what a
    released checkpoint's 'model' entry holds (xtrainer.py:517-529) - for a small geometry with a refiner."""
    from src.model.transformer import TransformerModel
    P = synth.model_params(64, 2, 2, 120, refine_layers=1)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True)
    sd = m.state_dict()
    keys = sorted(sd.keys())
    shapes = np.zeros((len(keys), 2), dtype=np.int64)
    for i, k in enumerate(keys):
        sh = tuple(sd[k].shape)
        shapes[i, :len(sh)] = sh
    np.savez_compressed(os.path.join(OUT, 'state_dict_enum.npz'), keys=np.array(keys), shapes=shapes,
                        geometry=np.array([64, 2, 2, 120, 1]))
    print('state_dict_enum.npz: %d entries' % len(keys))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'enum':
        gen_state_dict_enumeration()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'refiner':
        gen_refiner_goldens()
        sys.exit(0)
    gen_state_dict_enumeration()
    gen_refiner_goldens()
    gen_clcm_goldens()
    gen_region_head_goldens()
    gen_text_and_itm_goldens()
    gen_unit_goldens()
    gen_model_goldens()
    gen_trainer_goldens()
