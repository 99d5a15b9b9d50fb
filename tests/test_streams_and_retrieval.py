"""Remaining §8 rows on the GPU: the text-only crossfwd stream behind mlm_step (a14), the
sample_n = 4 relation loss of the t2i/i2t fine-tune steps (a13, BASELINE configs[4]) and the
retrieval scoring / Recall@K arithmetic (§8 f1)."""
import os

import numpy as np
import pytest
import torch

from m3p_amd import synth
from tests.util import rel_l2, max_abs

pytestmark = pytest.mark.gpu


def _build(cfg, extra=None):
    from m3p_amd.model.transformer import TransformerModel
    P = synth.model_params(cfg['emb_dim'], cfg['n_heads'], cfg['n_layers'], cfg['n_words'])
    for k, v in (extra or {}).items():
        setattr(P, k, v)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True)
    sd = synth.golden_state_dict(synth.hot_param_shapes(P))
    m.load_state_dict(sd, strict=False)
    return m.cuda(), P, sd


def test_crossfwd_text_stream_vs_reference_golden(golden_dir):
    g = dict(np.load(os.path.join(golden_dir, 'cfg1_text_itm.npz')))
    cfg = synth.CONFIGS['cfg1']
    m, P, sd = _build(cfg)
    m.eval()
    batch = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'])
    out = m('crossfwd', stream_='text', x=batch['x'].cuda(), lengths=batch['lengths'].cuda(), positions=None, langs=None,
            causal=False)
    assert out.shape == (cfg['T'], cfg['B'], cfg['emb_dim'])
    assert rel_l2(out.float(), g['text_out']) < 1e-2
    _, mlm = m('predict', tensor=out, pred_mask=batch['pred_mask'].cuda(), y=batch['y'].cuda(), get_scores=False)
    assert abs(float(mlm) - float(g['text_mlm_loss'])) < 5e-3


def test_mlm_step_on_batch_and_rel_steps(golden_dir):
    from m3p_amd.trainer import XTrainer
    from oracle import ref_cpu as O
    g = dict(np.load(os.path.join(golden_dir, 'cfg1_text_itm.npz')))
    cfg = synth.CONFIGS['cfg1']
    extra = dict(optimizer='adam_inverse_sqrt,beta1=0.9,beta2=0.98,lr=0.0001', clip_grad_norm=5, amp=-1, fp16=False,
                 accumulate_gradients=1, multi_gpu=False, epoch_size=100, cross_mlm_steps=[], cross_mrm_steps=[],
                 cross_mrfr_steps=[], cross_clcm_steps=[], sample_n=4, refine_image=False, multi_cls_loss_weight=1,
                 bin_cls_loss_weight=1, batch_size=cfg['B'], dump_path='/nonexistent_m3p_dump')
    m, P, sd = _build(cfg, extra)
    tr = XTrainer(m, {}, P)
    batch = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'])
    # text-only MLM step (Trainer.mlm_step loss path)
    loss = tr.mlm_step_on_batch(batch['x'], batch['lengths'], batch['pred_mask'], batch['y'])
    assert abs(float(loss) - float(g['text_mlm_loss'])) < 5e-3
    tr.iter()
    # t2i fine-tune step with sample_n = 4: CE over groups + BCE (xtrainer.py:1929-1938)
    m2, P2, _ = _build(cfg, extra)
    tr2 = XTrainer(m2, {}, P2)
    B, R = cfg['B'], cfg['R']
    img = batch['x_img'].transpose(0, 1).contiguous()
    loc = batch['image_loc'].transpose(0, 1).contiguous()
    pos = g['rel4_pos'].tolist()
    # the tuple retrieval_collate emits (xtrainer.py:1897): (sent, lengths, langs), (img, mask, loc, obj, pos, ids)
    tup = ((batch['x'], batch['lengths'], torch.zeros_like(batch['x'])),
           (img, torch.ones(B, R, dtype=torch.long), loc, torch.full((B, R), -1), pos, list(range(B))))
    loss = tr2.t2i_step(tup, 'coco', 1.0)
    assert abs(float(loss) - (float(g['rel4_ce']) + float(g['rel4_bce']))) < 1e-2
    assert tr2.stats['processed_s'] == B
    assert float(m2.arena().grad.abs().max()) == 0.0          # step taken, grads zeroed
    assert not m2.arena().touched
    # parameters never touched by the relation-only step keep their Adam step count at 0
    opt = tr2.optimizers['model']
    assert opt.state[m2.pred_layer.proj.bias]['step'] == 0
    assert opt.state[m2.seq_relationship.weight]['step'] == 1


def test_retrieval_scores_and_recall_vs_oracle():
    from m3p_amd import evaluation as E
    from oracle import ref_cpu as O
    cfg = dict(emb_dim=128, n_heads=4, n_layers=2, n_words=500, T=20, R=6)
    m, P, sd = _build(cfg)
    n_img, per = 6, 2
    n_cap = n_img * per
    b = synth.make_batch(cfg['T'], cfg['R'], n_cap, cfg['n_words'], 0, seed=21)
    bi = synth.make_batch(cfg['T'], cfg['R'], n_img, cfg['n_words'], 0, seed=22)
    scores, mine = E.relation_score_matrix(m, b['x'].cuda(), b['lengths'].cuda(), bi['x_img'].cuda(), bi['image_loc'].cuda(),
                                           chunk=5, img_block=4)
    assert scores.shape == (n_img, n_cap) and mine.tolist() == list(range(n_img))
    ref = torch.empty(n_img, n_cap)
    for i in range(n_img):
        xi = bi['x_img'][:, i:i + 1].expand(cfg['R'], n_cap, 2048)
        li = bi['image_loc'][:, i:i + 1].expand(cfg['R'], n_cap, 5)
        out = O.jointfwd(sd, cfg['n_layers'], cfg['n_heads'], b['x'], b['lengths'], xi, torch.full((n_cap,), cfg['R']), li)
        ref[i] = O.predict_relation(sd, out.transpose(0, 1)).view(-1)
    assert max_abs(scores, ref) < 3e-2
    # Recall@K arithmetic: identical when fed the same matrix; and agreeing between the two paths
    gt = torch.zeros(n_img, n_cap, dtype=torch.bool)
    for i in range(n_img):
        gt[i, i * per:(i + 1) * per] = True
    r_ref = E.recall_at_k(ref, gt, ks=(1, 5))
    r_or = O.recall_at_k(ref, gt.float().argmax(1), ks=(1,))
    assert 0.0 <= r_ref[1] <= r_ref[5] <= 1.0
    r_hip = E.recall_at_k(scores.cpu(), gt, ks=(1, 5))
    top1_same = float((scores.cpu().argmax(1) == ref.argmax(1)).float().mean())
    assert top1_same >= 0.8 and abs(r_hip[5] - r_ref[5]) <= 1.0 / n_img + 1e-9
    # sharded scoring (rank 1 of 2) covers the complementary images
    s1, mine1 = E.relation_score_matrix(m, b['x'].cuda(), b['lengths'].cuda(), bi['x_img'].cuda(), bi['image_loc'].cuda(), chunk=12, rank=1, world=2)
    assert mine1.tolist() == [1, 3, 5] and max_abs(s1, scores[mine1]) < 2e-2     # other tiling: bf16 noise only
    # both-direction metric of the evaluator (xevaluator.py:1621-1657): vectorised == the oracle's loops, exactly
    assert E.retrieval_recalls(scores, gt.float()) == O.retrieval_recalls(scores.cpu(), gt.float())
    assert E.retrieval_recalls(ref.cuda(), gt.float()) == O.retrieval_recalls(ref, gt.float())


def _rel_params(cfg, sample_n):
    return dict(optimizer='adam_inverse_sqrt,beta1=0.9,beta2=0.98,lr=0.0001', clip_grad_norm=5, amp=1, fp16=True,
                accumulate_gradients=1, multi_gpu=False, epoch_size=100, cross_mlm_steps=[], cross_mrm_steps=[],
                cross_mrfr_steps=[], cross_clcm_steps=[], sample_n=sample_n, refine_image=False, multi_cls_loss_weight=1,
                bin_cls_loss_weight=1, batch_size=cfg['B'] // sample_n, dump_path='/nonexistent_m3p_dump')


def _rel_tuple(batch, sl, R, pos):
    n = sl.stop - sl.start
    x = batch['x'][:, sl].contiguous()
    return ((x, batch['lengths'][sl].contiguous(), torch.zeros_like(x)),
            (batch['x_img'][:, sl].transpose(0, 1).contiguous(), torch.ones(n, R, dtype=torch.long),
             batch['image_loc'][:, sl].transpose(0, 1).contiguous(), torch.full((n, R), -1), pos, list(range(n))))


def test_cfg5_geometry_finetune_steps_vs_oracle():
    """BASELINE configs[4] geometry (768d / 12 heads, 80 tokens + 36 regions, groups of sample_n = 4) at two layers and a
    small vocabulary, where the oracle runs in seconds: t2i_step / i2t_step losses (CE over groups + BCE) and the
    gradients the step feeds the optimizer, against the oracle."""
    from m3p_amd.trainer import XTrainer
    from oracle import ref_cpu as O
    cfg = dict(emb_dim=768, n_heads=12, n_layers=2, n_words=5000, T=80, R=36, B=8, n_pred=0)
    m, P, sd = _build(cfg, _rel_params(cfg, 4))
    tr = XTrainer(m, {}, P)
    batch = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], 0, seed=5, ragged=True)
    pos = [3, 1]
    grads = {}
    opt = tr.optimizers['model']
    inner = opt.step

    def step(closure=None):
        torch.cuda.synchronize()
        grads.update({n: p.grad.detach().float().cpu().clone() for n, p in m.named_parameters() if getattr(p, '_m3p_arena', None)})
        return inner(closure)
    opt.step = step
    loss = tr.i2t_step(_rel_tuple(batch, slice(0, cfg['B']), cfg['R'], pos), 'flicker', 1.0)
    names = list(sd.keys())
    leaves = {n: sd[n].clone().requires_grad_(True) for n in names}
    batch['pos_labels'] = torch.tensor(pos)
    batch['pred_mask'] = torch.zeros_like(batch['x'], dtype=torch.bool)
    res = O.pretrain_losses(leaves, cfg['n_layers'], cfg['n_heads'], batch, cfg['R'], sample_n=4, multi_w=1.0, bin_w=1.0)
    assert abs(float(loss) - float(res['itm'])) < 5e-3
    used = [n for n in names if n not in ('pred_layer.proj.bias',)]
    gref = dict(zip(used, torch.autograd.grad(res['total'], [leaves[n] for n in used], allow_unused=True)))
    qb = float(gref['attentions.0.q_lin.bias'].norm())
    bad = []
    for n in used:
        if gref[n] is None:
            continue
        if '.k_lin.bias' in n:
            assert float(grads[n].norm()) < 5e-2 * qb + 1e-6, n
        elif rel_l2(grads[n], gref[n]) > 5e-2:
            bad.append((n, rel_l2(grads[n], gref[n])))
    assert not bad, bad
    assert tr.stats['processed_s'] == cfg['B'] and tr.stats['processed_w'] == cfg['B'] * (cfg['T'] + cfg['R'])


def test_cfg5_full_size_finetune_step_properties():
    """configs[4] at its workload: 12L / 768d / V = 250 002, 80 + 36, 24 items x sample_n 4 = 96 sequences per step.
    The oracle cannot run this in seconds; size-independent properties instead: the 96-sequence relation loss and its
    gradients are the average of the two 48-sequence halves (12 groups each); the first step's loss is the loss of a
    near-uniform guess, ln 4 + ln 2-ish; three optimizer steps stay finite and lower the loss on the same batch."""
    import math
    from m3p_amd.model.transformer import TransformerModel
    from m3p_amd.trainer import XTrainer
    cfg = synth.CONFIGS['cfg5']
    P = synth.model_params(cfg['emb_dim'], cfg['n_heads'], cfg['n_layers'], cfg['n_words'])
    for k, v in _rel_params(cfg, 4).items():
        setattr(P, k, v)
    torch.manual_seed(1234)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True).cuda()
    tr = XTrainer(m, {}, P)
    B, R = cfg['B'], cfg['R']
    batch = synth.make_batch(cfg['T'], R, B, cfg['n_words'], 0, seed=9, ragged=True)
    pos = [int(v) for v in np.random.RandomState(3).randint(0, 4, size=B // 4)]
    names = ['attentions.0.q_lin.weight', 'attentions.11.out_lin.weight', 'ffns.5.lin1.weight', 'ffns.11.lin2.bias',
             'layer_norm2.7.weight', 'image_embeddings.image_embeddings.weight', 'pooled_layer.dense.weight',
             'seq_relationship.weight', 'position_embeddings.weight']
    own = dict(m.named_parameters())
    tr.params.clip_grad_norm = 0

    def grads_of(sl, pos_sl):
        m.arena().zero_grad()
        tr.n_iter = 1
        tr.params.accumulate_gradients = 2           # non-boundary micro-step: backward only, nothing moves
        loss = tr.t2i_step(_rel_tuple(batch, sl, R, pos_sl), 'flicker', 1.0)
        torch.cuda.synchronize()
        return float(loss), {n: own[n].grad.detach().float().clone() for n in names}

    full, g = grads_of(slice(0, B), pos)
    la, ga = grads_of(slice(0, B // 2), pos[:B // 8])
    lb, gb = grads_of(slice(B // 2, B), pos[B // 8:])
    assert abs(0.5 * (la + lb) - full) < 2e-3
    assert 0.8 * (math.log(4) + math.log(2)) < full < 1.6 * (math.log(4) + math.log(2)), full
    bad = [(n, rel_l2(0.5 * (ga[n] + gb[n]), g[n])) for n in names]
    bad = [(n, e) for n, e in bad if e > 2e-2]
    assert not bad, bad
    # real steps
    m.arena().zero_grad()
    tr.params.accumulate_gradients, tr.params.clip_grad_norm, tr.n_iter = 1, 5, 0
    for g_ in tr.optimizers['model'].param_groups:
        g_['lr'] = 2e-5
    losses = []
    for _ in range(4):
        losses.append(float(tr.i2t_step(_rel_tuple(batch, slice(0, B), R, pos), 'flicker', 1.0)))
        for g_ in tr.optimizers['model'].param_groups:
            g_['lr'] = 2e-5
        tr.n_iter += 1
    assert all(math.isfinite(v) for v in losses) and losses[-1] < losses[0], losses
    assert torch.isfinite(m.arena().master).all()


def test_cfg5_retrieval_1000x5_shard_and_oracle_subset():
    """Retrieval evaluation at configs[4] scale (xevaluator.py:1528-1657): a synthetic Multi30K-shaped test set of 1000
    images x 5 captions, T = 80, R = 36, on the 12L / 768d / V = 250 002 model.  One image shard of the exhaustive
    1000 x 5000 scoring (stride sharding, as 25 ranks would split it: 40 images x 5000 captions = 200 000 encoder
    passes) is scored on the MI355X; the metric read off it equals the oracle's metric loops on the same matrix
    exactly, in both directions; and on a sub-block the CPU oracle can score (the full 12 layers), scores agree within
    the bf16 bar, every query whose oracle margin exceeds twice that bar has the identical top-1, and Recall@1 of
    the two sub-blocks differs by no more than the near-tie queries."""
    from m3p_amd import evaluation as E
    from m3p_amd.model.transformer import TransformerModel
    from oracle import ref_cpu as O
    cfg = synth.CONFIGS['cfg5']
    n_img, per = 1000, 5
    n_cap = n_img * per
    P = synth.model_params(cfg['emb_dim'], cfg['n_heads'], cfg['n_layers'], cfg['n_words'])
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True)
    sd = synth.golden_state_dict(synth.hot_param_shapes(P), scale=0.05)
    m.load_state_dict(sd, strict=False)
    m = m.cuda()
    rs = np.random.RandomState(2024)
    T, R = cfg['T'], cfg['R']
    lengths = torch.from_numpy(rs.randint(8, T + 1, size=n_cap).astype(np.int64))
    lengths[0] = T
    x = torch.full((T, n_cap), synth.PAD, dtype=torch.long)
    toks = torch.from_numpy(rs.randint(4, cfg['n_words'] - 1, size=(T, n_cap)).astype(np.int64))
    alive = torch.arange(T)[:, None] < lengths[None, :]
    x[alive] = toks[alive]
    x[0] = synth.BOS
    x[lengths - 1, torch.arange(n_cap)] = synth.EOS
    feats = rs.standard_normal((R, n_img, 2048)).astype(np.float32)
    feats /= np.linalg.norm(feats, axis=-1, keepdims=True)
    loc = rs.uniform(size=(R, n_img, 5)).astype(np.float32)
    loc /= np.linalg.norm(loc, axis=-1, keepdims=True)
    x_img, image_loc = torch.from_numpy(feats), torch.from_numpy(loc)
    labels = torch.zeros(n_img, n_cap)
    for i in range(n_img):
        labels[i, i * per:(i + 1) * per] = 1
    world, rank = 25, 3
    scores, mine = E.relation_score_matrix(m, x.cuda(), lengths.cuda(), x_img.cuda(), image_loc.cuda(), chunk=500, img_block=4,
                                           rank=rank, world=world)
    torch.cuda.synchronize()
    assert scores.shape == (n_img // world, n_cap) and mine.tolist() == list(range(rank, n_img, world))
    assert torch.isfinite(scores).all()
    lab_shard = labels[mine.cpu()]
    got = E.retrieval_recalls(scores, lab_shard)
    assert got == O.retrieval_recalls(scores.cpu(), lab_shard)
    assert all(0.0 <= v <= 1.0 for v in got)
    # the oracle on a 3-image x 40-caption block of that shard (all 12 layers, fp32 CPU)
    ii = mine.cpu()[:3]
    cc = torch.cat([torch.arange(int(i) * per, int(i) * per + per) for i in ii] + [torch.arange(2000, 2025)])
    ref = torch.empty(len(ii), len(cc))
    with torch.no_grad():
        for r, i in enumerate(ii.tolist()):
            xi = x_img[:, i:i + 1].expand(R, len(cc), 2048)
            li = image_loc[:, i:i + 1].expand(R, len(cc), 5)
            out = O.jointfwd(sd, cfg['n_layers'], cfg['n_heads'], x[:, cc], lengths[cc], xi, torch.full((len(cc),), R), li)
            ref[r] = O.predict_relation(sd, out.transpose(0, 1)).view(-1)
    sub = scores[:3].cpu()[:, cc]
    tol = 3e-2
    assert max_abs(sub, ref) < tol, max_abs(sub, ref)
    top2 = ref.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 2 * tol
    assert bool((sub.argmax(1)[clear] == ref.argmax(1)[clear]).all())
    sub_lab = labels[ii][:, cc]
    r_hip, r_ref = E.retrieval_recalls(sub, sub_lab), O.retrieval_recalls(ref, sub_lab)
    assert abs(r_hip[3] - r_ref[3]) <= float((~clear).sum()) / len(ii) + 1e-9
    # every HIP top-1 is within the bar of the oracle's best score for that query (ranking equal up to the noise floor)
    best_ref = ref.max(dim=1).values
    picked = ref[torch.arange(len(ii)), sub.argmax(1)]
    assert bool(((best_ref - picked) <= 2 * tol).all())


def test_trainer_entry_points_track_the_reference_run(golden_dir):
    """mlm_step('en', None, 1) on a monolingual stream (generate_batch -> round_batch -> mask_out -> crossfwd -> predict ->
    optimize) and t2i_step / i2t_step on the tuple retrieval_collate emits, against the same calls made on the
    reference's own XTrainer on CPU (tests/golden/host_logic.npz): losses, learning rate, word / sentence counters and
    parameter norms after the steps.  Same seeds => the very same masked batch (tests/test_host_surface.py pins that)."""
    from m3p_amd.trainer import XTrainer
    G = np.load(os.path.join(golden_dir, 'host_logic.npz'))
    cfg = synth.CONFIGS['cfg1']
    common = dict(optimizer='adam_inverse_sqrt,beta1=0.9,beta2=0.98,lr=0.0001', clip_grad_norm=5, amp=-1, fp16=False,
                  accumulate_gradients=1, multi_gpu=False, epoch_size=100, cross_mlm_steps=[], cross_mrm_steps=[],
                  cross_mrfr_steps=[], cross_clcm_steps=[], cross_rel_steps=[('google', 'img')], refine_image=False,
                  batch_size=cfg['B'], dump_path='/nonexistent_m3p_dump', langs=['en'], sample_alpha=0, word_pred=0.15,
                  word_mask=0.8, word_keep=0.1, word_rand=0.1, t2i_flag=True, i2t_flag=True)

    class _Stream:
        def __init__(self, batches):
            self.batches = batches

        def get_iterator(self, shuffle=True):
            return iter(self.batches)

    batch = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], 0)
    m, P, sd = _build(cfg, dict(common, sample_n=2, multi_cls_loss_weight=0, bin_cls_loss_weight=1))
    tr = XTrainer(m, {'mono_stream': {'en': {'train': _Stream([(batch['x'], batch['lengths'])])}}}, P)
    np.random.seed(77); torch.manual_seed(77)
    loss = tr.mlm_step('en', None, 1.0)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(G['mlm_step_loss'])) < 5e-3
    assert abs(tr.optimizers['model'].param_groups[0]['lr'] - float(G['mlm_step_lr'])) < 1e-15
    assert int(torch.stack(tr._pending_w).sum()) == int(G['mlm_step_processed_w'])      # the same number of masked words
    assert tr.stats['processed_s'] == cfg['B'] and tr.n_sentences == cfg['B']
    own = dict(m.named_parameters())
    for k in ('embeddings.weight', 'attentions.0.q_lin.weight', 'layer_norm2.1.weight', 'pred_layer.proj.bias'):
        assert abs(float(own[k].detach().norm()) - float(G['mlm_step_pnorm/' + k])) < 1e-4 * float(G['mlm_step_pnorm/' + k]), k
    with pytest.raises(KeyError):
        tr.mlm_step('en', 'en', 1.0)            # the lang1 == lang2 batch reads data['mono'] (xtrainer.py:497-501): none here
    assert tr.mlm_step('en', None, 0) is None   # lambda 0: nothing happens (xtrainer.py:740-741)

    m3, P3, _ = _build(cfg, dict(common, sample_n=4, multi_cls_loss_weight=1, bin_cls_loss_weight=1))
    tr3 = XTrainer(m3, {}, P3)
    B, R = cfg['B'], cfg['R']
    tup = [(batch['x'], batch['lengths'], torch.zeros_like(batch['x'])),
           [batch['x_img'].transpose(0, 1).contiguous(), torch.ones(B, R, dtype=torch.long),
            batch['image_loc'].transpose(0, 1).contiguous(), torch.full((B, R), -1, dtype=torch.long), [2, 0], list(range(B))]]
    l1 = tr3.t2i_step(tup, 'google', 1.0)
    l2 = tr3.i2t_step(tup, 'google', 0.5)
    torch.cuda.synchronize()
    assert abs(float(l1) - float(G['t2i_step_loss'])) < 1e-2 and abs(float(l2) - float(G['i2t_step_loss'])) < 1e-2
    ps, pw, ns = (int(v) for v in G['rel_step_processed'])
    assert (tr3.stats['processed_s'], tr3.stats['processed_w'], tr3.n_sentences) == (ps, pw, ns)
    own = dict(m3.named_parameters())
    for k in ('pooled_layer.dense.weight', 'attentions.1.out_lin.weight', 'embeddings.weight'):
        assert abs(float(own[k].detach().norm()) - float(G['rel_step_pnorm/' + k])) < 1e-4 * float(G['rel_step_pnorm/' + k]), k
    assert tr3.t2i_step(tup, 'google', 0) is None


def test_pretrain_rel_step_and_rel_step_through_a_dataloader():
    """pretrain_rel_step / rel_step (xtrainer.py:1867-1886): get_batch -> DataLoader(dataset, collate) -> the two task
    steps.  A synthetic dataset of per-item tuples stands in for the reference's HDF5 readers."""
    from m3p_amd.trainer import XTrainer
    cfg = dict(emb_dim=128, n_heads=4, n_layers=2, n_words=1000, T=16, R=10, B=4, n_pred=0)
    k, R, V = 2, cfg['R'], cfg['n_words']
    rs = np.random.RandomState(4)

    def item(pretrain, i2t):
        caps = [rs.randint(4, V - 1, size=rs.randint(3, 12)).astype(np.int64) for _ in range(k)]
        feats = torch.from_numpy(rs.standard_normal((k, R, 2048)).astype(np.float32))
        feats = feats / feats.norm(dim=-1, keepdim=True)
        masks = torch.ones(k, R, dtype=torch.long)
        boxes = torch.from_numpy(rs.uniform(size=(k, R, 5)).astype(np.float32))
        objs = torch.from_numpy(np.where(rs.rand(k, R) < 0.3, rs.randint(0, 1600, size=(k, R)), -1).astype(np.int64))
        objs[0, 0] = 5
        ids = [int(v) for v in rs.randint(0, 1000, size=k)]
        if not pretrain:
            return (caps, feats, masks, boxes, objs, [int(rs.randint(0, k))], ids, [0] * k)
        lm = [[int(w) if j == 1 else -1 for j, w in enumerate(c)] for c in caps]
        base = (caps, feats, masks, boxes, objs, lm, int(rs.randint(0, k)), ids, feats.clone(), [0] * k)
        if not i2t:
            return base
        caps2 = [rs.randint(4, V - 1, size=rs.randint(3, 9)).astype(np.int64) for _ in range(k)]
        return base + (caps2, torch.from_numpy(rs.randint(0, 2, size=k).astype(np.int64)))

    for pretrain in (True, False):
        data = [(item(pretrain, False), item(pretrain, True)) for _ in range(6)]
        extra = dict(optimizer='adam_inverse_sqrt,beta1=0.9,beta2=0.98,lr=0.0001', clip_grad_norm=5, amp=1, fp16=True,
                     accumulate_gradients=1, multi_gpu=False, epoch_size=100, batch_size=2, dump_path='/nonexistent_m3p_dump',
                     cross_rel_steps=[('coco', 'img')], cross_mlm_steps=[('coco', 'img')] if pretrain else [],
                     cross_mrm_steps=[('coco', 'img')] if pretrain else [], cross_mrfr_steps=[('coco', 'img')] if pretrain else [],
                     cross_clcm_steps=[('coco', 'img')] if pretrain else [], sample_n=k, refine_image=False,
                     multi_cls_loss_weight=1, bin_cls_loss_weight=1, is_pretrain=pretrain, n_gpu_per_node=1, num_workers=0,
                     t2i_flag=True, i2t_flag=True, lambda_t2i='1', lambda_i2t='1', lambda_mlm='1', lambda_mrm='1', lambda_mrfr='1')
        m, P, sd = _build(cfg, extra)
        tr = XTrainer(m, {'cross_modal': {('coco', 'img'): {'train': data}}}, P)
        before = m.arena().master.clone()
        for _ in range(4):                      # 3 batches per epoch: the fourth call re-creates the iterator
            if pretrain:
                tr.pretrain_rel_step('coco', 'img')
            else:
                tr.rel_step('coco', 'img', 1.0, 1.0)
            tr.iter()
        torch.cuda.synchronize()
        assert tr.n_sentences == 4 * 2 * P.batch_size and tr.stats['processed_s'] == 4 * 2 * 2 * k
        assert torch.isfinite(m.arena().master).all() and not torch.equal(before, m.arena().master)
        keys = {'t2i-coco', 'i2t-coco'} | ({'CMLM-coco', 'MRM-coco', 'MRFR-coco', 'CLCM-coco'} if pretrain else set())
        assert all(len(tr.stats[s]) > 0 for s in keys), {s: len(tr.stats.get(s, [])) for s in keys}


def test_text_stream_with_language_embeddings_vs_reference(golden_dir):
    """The text MLM stream of a multilingual model (n_langs = 2; sentence b in language b % 2): crossfwd adds
    cross_lang_embeddings(langs) (transformer.py:1059-1060).  Output, loss and gradients - the language table's and the
    vocabulary rows' included - against the reference's (tests/golden/text_langs.npz) and the oracle."""
    from m3p_amd.model.transformer import TransformerModel
    from oracle import ref_cpu as O
    g = dict(np.load(os.path.join(golden_dir, 'text_langs.npz')))
    cfg, P, sd, batch, langs = synth.text_langs_case()
    o = O.crossfwd_text(sd, cfg['n_layers'], cfg['n_heads'], batch['x'], batch['lengths'], langs=langs)
    assert float((o - torch.from_numpy(g['text_out'])).abs().max()) < 1e-4
    torch.manual_seed(0)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True).cuda()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and 'cross_lang_embeddings.weight' not in missing
    m.train()
    m.arena().zero_grad()
    out = m('crossfwd', stream_='text', x=batch['x'].cuda(), lengths=batch['lengths'].cuda(), positions=None,
            langs=langs.cuda(), causal=False)
    assert rel_l2(out.float(), g['text_out']) < 1e-2
    _, loss = m('predict', tensor=out, pred_mask=batch['pred_mask'].cuda(), y=batch['y'].cuda(), get_scores=False)
    assert abs(float(loss) - float(g['mlm_loss'])) < 5e-3
    loss.backward()
    own = dict(m.named_parameters())
    for k in ('cross_lang_embeddings.weight', 'position_embeddings.weight', 'layer_norm_emb.weight', 'attentions.0.q_lin.weight',
              'ffns.1.lin2.weight', 'pred_layer.proj.bias'):
        assert rel_l2(own[k].grad.float(), g['grad.' + k]) < 3e-2, k
    ge = own['embeddings.weight'].grad.float()
    ids = torch.from_numpy(g['grad_rows.ids']).cuda()
    assert rel_l2(ge[ids], g['grad_rows.embeddings.weight']) < 3e-2
    assert abs(float(ge.norm()) - float(g['grad_norm.embeddings.weight'])) < 3e-2 * float(g['grad_norm.embeddings.weight'])
    assert 'cross_lang_embeddings.weight' in m.arena().touched
    # without language ids the same model runs the plain stream; ids on a one-language model are refused
    out0 = m('crossfwd', stream_='text', x=batch['x'].cuda(), lengths=batch['lengths'].cuda(), causal=False)
    assert rel_l2(out0.float(), O.crossfwd_text(sd, cfg['n_layers'], cfg['n_heads'], batch['x'], batch['lengths'])) < 1e-2


def test_freelb_t2i_step_tracks_the_reference_run(golden_dir):
    """freelb_t2i_step (xtrainer.py:2021-2121): three adversarial passes over one batch - perturbed word embeddings through
    jointfwd(text_embed=) and perturbed region features, each pass an optimizer step, a normalised ascent step on both
    perturbations in between - against the same call on the reference's XTrainer on CPU under the same torch seed
    (tests/golden/freelb_step.npz): the summed loss, the schedule's state and how far the parameters moved (Adam steps of
    1e-7-scale learning rates: the displacement, not the norm, is what three steps change)."""
    from m3p_amd.trainer import XTrainer
    G = np.load(os.path.join(golden_dir, 'freelb_step.npz'))
    cfg = synth.CONFIGS['cfg1']
    common = dict(optimizer='adam_inverse_sqrt,beta1=0.9,beta2=0.98,lr=0.0001', clip_grad_norm=5, amp=-1, fp16=False,
                  accumulate_gradients=1, multi_gpu=False, epoch_size=100, cross_mlm_steps=[], cross_mrm_steps=[],
                  cross_mrfr_steps=[], cross_clcm_steps=[], cross_rel_steps=[('google', 'img')], refine_image=False,
                  batch_size=cfg['B'], dump_path='/nonexistent_m3p_dump', langs=['en'], t2i_flag=True, i2t_flag=True, is_freelb=True)
    batch = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], 0)
    m, P, _ = _build(cfg, dict(common, sample_n=4, multi_cls_loss_weight=1, bin_cls_loss_weight=1))
    tr = XTrainer(m, {}, P)
    B, R = cfg['B'], cfg['R']
    tup = [(batch['x'], batch['lengths'], torch.zeros_like(batch['x'])),
           [batch['x_img'].transpose(0, 1).contiguous(), torch.ones(B, R, dtype=torch.long),
            batch['image_loc'].transpose(0, 1).contiguous(), torch.full((B, R), -1, dtype=torch.long), [2, 0], list(range(B))]]
    own = dict(m.named_parameters())
    before = {k: own[k].detach().clone() for k in ('embeddings.weight', 'attentions.0.q_lin.weight', 'pooled_layer.dense.weight')}
    torch.manual_seed(4242)
    loss = tr.freelb_t2i_step(tup, 'google', 1.0)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(G['loss'])) < 1e-2
    group = tr.optimizers['model'].param_groups[0]
    assert group['num_updates'] == int(G['n_updates']) == 3 and abs(group['lr'] - float(G['lr'])) < 1e-15
    for k, b in before.items():
        moved = float((own[k].detach() - b).norm())
        assert abs(moved - float(G['dnorm/' + k])) < 0.1 * float(G['dnorm/' + k]), (k, moved, float(G['dnorm/' + k]))
    rows = int(((own['embeddings.weight'].detach() - before['embeddings.weight']).abs().sum(1) > 0).sum())
    assert rows == int(G['emb_rows_moved'])          # the word-embedding gradient of text_embed reached the matrix, for the batch's words only
    assert 'FRLB-t2i-google' in tr.stats and tr.stats['processed_s'] == B
    assert tr.freelb_i2t_step(tup, 'google', 0) is None
