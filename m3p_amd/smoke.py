"""One small invocation of the hot path on cuda:0, checked against the oracle
(called by __graft_entry__.smoke())."""
import torch

from . import synth


def run():
    from oracle import ref_cpu as O   # the checker, never the thing measured
    from .model.transformer import TransformerModel
    cfg = synth.CONFIGS['cfg1']
    P = synth.model_params(cfg['emb_dim'], cfg['n_heads'], cfg['n_layers'], cfg['n_words'])
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True)
    sd = synth.golden_state_dict(synth.hot_param_shapes(P))
    m.load_state_dict(sd, strict=False)
    m = m.cuda().train()
    batch = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'])
    dev = 'cuda:0'
    out = m('jointfwd', x=batch['x'].to(dev), lengths=batch['lengths'].to(dev), x_img=batch['x_img'].to(dev),
            lengths_img=batch['lengths_img'].to(dev), causal=False, langs=None, image_loc=batch['image_loc'].to(dev),
            refine_image=False)
    _, mlm = m('predict', tensor=out[cfg['R']:], pred_mask=batch['pred_mask'].to(dev), y=batch['y'].to(dev), get_scores=False)
    rel = m('predict', tensor=out.transpose(0, 1), is_relation=True)
    onehot = torch.eye(2, device=dev)[batch['pos_labels'].to(dev)].reshape(-1)
    bce = torch.nn.functional.binary_cross_entropy_with_logits(rel.view(-1).float(), onehot)
    (mlm + bce).backward()
    torch.cuda.synchronize()
    ref = O.pretrain_losses(sd, cfg['n_layers'], cfg['n_heads'], batch, cfg['R'])
    err_out = float((out.float().cpu() - ref['out']).norm() / ref['out'].norm())
    assert err_out < 1e-2, 'encoder output relL2 %.3e vs oracle' % err_out
    assert abs(float(mlm) - float(ref['mlm'])) < 5e-3, (float(mlm), float(ref['mlm']))
    assert abs(float(bce) - float(ref['itm'])) < 5e-3
    gnorm = float(m.arena().grad.norm())
    assert gnorm > 0 and gnorm == gnorm
    print('smoke ok: out relL2 %.2e, mlm %.4f (oracle %.4f), itm %.4f (oracle %.4f), |grad| %.4f'
          % (err_out, float(mlm), float(ref['mlm']), float(bce), float(ref['itm']), gnorm))
