"""End-to-end parity of the MI355X model against the oracle and the reference's golden
vectors (BASELINE.json configs[0] = cfg1, plus a ragged d=768/dh=64 case).  bf16 bars from
SURVEY.md §8c: output relL2 <= 1e-2, loss <= 5e-3, grads relL2 <= 5e-2,
attentions.*.k_lin.bias absolute (its true gradient is 0)."""
import os

import numpy as np
import pytest
import torch

from m3p_amd import synth
from tests.util import encoder_keep_masks, rel_l2, max_abs

pytestmark = pytest.mark.gpu

# 'mid': the cfg2 width at a short ragged sequence; 'large': the geometry of BASELINE configs[3] (M3P-large: d=1024,
# 16 heads, 100 regions + 256 tokens = 356 keys -> the 12-step attention instantiation) at 2 layers / small V;
# 'tiny': smallest shapes: one region, six tokens, two sequences, head dim 32, one layer, one MLM target each.
# 'tiles' / 'tiles768' (round 5): sizes at which EVERY whole-tile branch of the benchmarked dispatch fires - M = B * S and the
# number of predicted rows are multiples of 256 (M >= 1024, n = 4096): lin1 + GELU + one-byte gelu' epilogue, byte-decoding dU
# epilogue, vocabulary projection with block statistics, vocabulary data gradient on the four-wave (tile, K-chunk) kernel, paired
# out_lin + q/k/v weight gradient, and - 'tiles768' only: cfg2's own layer geometry (d = 768, 12 heads, 36 + 128 positions: the
# 6-step / 11-tile attention instantiation, K = 3072 products on the four-wave NT kernel) with a vocabulary of >= 1024 output
# tiles, so that the tied matrix's weight gradient is dealt out as whole tiles and STORED instead of accumulated.
CFGS = {'mid': dict(emb_dim=768, n_heads=12, n_layers=2, n_words=5000, T=40, R=36, B=6, n_pred=6),
        'large': dict(emb_dim=1024, n_heads=16, n_layers=2, n_words=5000, T=256, R=100, B=4, n_pred=38),
        'tiny': dict(emb_dim=128, n_heads=4, n_layers=1, n_words=300, T=6, R=1, B=2, n_pred=1),
        'tiles': dict(emb_dim=256, n_heads=4, n_layers=2, n_words=5000, T=48, R=16, B=256, n_pred=16),
        'tiles768': dict(emb_dim=768, n_heads=12, n_layers=1, n_words=88000, T=128, R=36, B=128, n_pred=32)}


def _cfg(name):
    return synth.CONFIGS['cfg1'] if name == 'cfg1' else CFGS[name]


class _DispatchSpy:
    """Records which launches the step took: epilogues handed to ops.gemm_nt, calls of the stream-K / paired / stored forms,
    and the return codes of the library's store entry point."""

    def __init__(self, monkeypatch):
        from m3p_amd import lib as L, ops
        self.epilogues, self.nt_shapes, self.wgrad_zero, self.store_rc = [], [], [], []
        self.streamk = self.pairs = 0
        real_nt, real_sk, real_pair, real_wg = ops.gemm_nt, ops.gemm_nn_streamk, ops.gemm_wgrad_pair, ops.gemm_wgrad
        lib = L.load()
        real_store = lib.m3p_gemm_wgrad_store_bf16

        def nt(a, w, epilogue=0, **kw):
            self.epilogues.append(epilogue)
            self.nt_shapes.append((epilogue, a.shape[0], kw.get('n') or w.shape[0], a.shape[1]))
            return real_nt(a, w, epilogue, **kw)

        def sk(*a, **kw):
            self.streamk += 1
            return real_sk(*a, **kw)

        def pair(*a, **kw):
            self.pairs += 1
            return real_pair(*a, **kw)

        def wg(*a, **kw):
            self.wgrad_zero.append(bool(kw.get('dw_is_zero')))
            return real_wg(*a, **kw)

        def store(*a):
            rc = real_store(*a)
            self.store_rc.append(rc)
            return rc
        monkeypatch.setattr(ops, 'gemm_nt', nt)
        monkeypatch.setattr(ops, 'gemm_nn_streamk', sk)
        monkeypatch.setattr(ops, 'gemm_wgrad_pair', pair)
        monkeypatch.setattr(ops, 'gemm_wgrad', wg)
        monkeypatch.setattr(lib, 'm3p_gemm_wgrad_store_bf16', store)

    def assert_benchmarked_dispatch(self, cfg, m):
        """Every branch `bench.py`'s cfg2 / cfg3 step takes was taken here too."""
        from m3p_amd import lib as L
        lib = L.load()
        d, M = cfg['emb_dim'], cfg['B'] * (cfg['T'] + cfg['R'])
        n, Vp = cfg['B'] * cfg['n_pred'], m.arena().V_pad
        assert M % 256 == 0 and n % 256 == 0 and n >= 4096
        for e in (L.EPI_BIAS_GELUQ, L.EPI_MULQ, L.EPI_BIAS_LSE):
            assert e in self.epilogues, (e, sorted(set(self.epilogues)))
        assert L.EPI_DGELU not in self.epilogues and L.EPI_BIAS_GELU not in self.epilogues
        assert (L.EPI_BIAS_GELUQ, M, 4 * d, d) in self.nt_shapes and (L.EPI_MULQ, M, 4 * d, d) in self.nt_shapes
        assert (L.EPI_BIAS_LSE, n, Vp, d) in self.nt_shapes
        # the kernels behind them (the library's own dispatch table, csrc/gemm.hip: nt_plan / wgrad_plan)
        assert lib.m3p_gemm_nt_plan(M, 4 * d, d, L.EPI_BIAS_GELUQ) == L.KERN_NT_W8
        assert lib.m3p_gemm_nt_plan(M, 3 * d, d, L.EPI_BIAS) == L.KERN_NT_W8
        assert lib.m3p_gemm_nt_plan(n, Vp, d, L.EPI_BIAS_LSE) == L.KERN_NT_W8
        if (4 * d // 256) * (d // 256) >= 9:       # (d = 256: four output tiles - the ring kernel's; tiles768 has cfg2's 36)
            assert lib.m3p_gemm_wgrad_plan(M, 4 * d, d) == L.KERN_WGRAD_W4_CHUNKS
            assert lib.m3p_gemm_wgrad_plan(M, d, 4 * d) == L.KERN_WGRAD_W4_CHUNKS
        assert self.streamk == 0, 'the vocabulary data gradient fell back to stream-K'
        assert self.pairs == cfg['n_layers'], 'out_lin + q/k/v weight gradients were not paired'
        assert True in self.wgrad_zero, 'the tied matrix was not known to be zero when its weight gradient ran'
        if 4 * d >= 2048:
            assert lib.m3p_gemm_nt_plan(M, d, 4 * d, L.EPI_BIAS_DROP_RES) == L.KERN_NT_W4
            assert lib.m3p_gemm_nt_plan(M, d, 4 * d, L.EPI_RES) == L.KERN_NT_W4
        if lib.m3p_gemm_wgrad_plan(n, Vp, d) == L.KERN_WGRAD_W4_TILES:
            assert self.store_rc == [0], self.store_rc           # stored, not accumulated
        else:
            assert self.store_rc == [-2], self.store_rc          # (tile, M-chunk) form: accumulates through the workspace


def _build(cfg, dropout=0.0):
    from m3p_amd.model.transformer import TransformerModel
    P = synth.model_params(cfg['emb_dim'], cfg['n_heads'], cfg['n_layers'], cfg['n_words'], dropout=dropout,
                           attention_dropout=dropout)
    torch.manual_seed(0)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True)
    sd = synth.golden_state_dict(synth.hot_param_shapes(P))
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected
    return m.cuda(), P, sd


def _losses(m, batch, R, sample_n=2):
    dev = 'cuda'
    out = m('jointfwd', x=batch['x'].to(dev), lengths=batch['lengths'].to(dev), x_img=batch['x_img'].to(dev),
            lengths_img=batch['lengths_img'].to(dev), causal=False, langs=None, image_loc=batch['image_loc'].to(dev),
            refine_image=False)
    _, mlm = m('predict', tensor=out[R:], pred_mask=batch['pred_mask'].to(dev), y=batch['y'].to(dev), get_scores=False)
    rel = m('predict', tensor=out.transpose(0, 1), is_relation=True)
    onehot = torch.eye(sample_n, device=dev)[batch['pos_labels'].to(dev)].reshape(-1)
    bce = torch.nn.functional.binary_cross_entropy_with_logits(rel.view(-1).float(), onehot)
    return out, mlm, rel, bce


def test_state_dict_names_match_reference_enumeration():
    """Every hot parameter name/shape of SURVEY §8b exists, the vocabulary projection is tied."""
    m, P, sd = _build(synth.CONFIGS['cfg1'])
    own = m.state_dict()
    for k, v in sd.items():
        assert tuple(own[k].shape) == tuple(v.shape), k
    assert own['pred_layer.proj.weight'].data_ptr() == own['embeddings.weight'].data_ptr()
    assert m.pred_layer.proj.weight is m.embeddings.weight
    for k in ('encoder_attn.0.q_lin.weight', 'layer_norm15.1.bias', 'mrfr_dense.weight', 'pred_obj_layer.proj.bias',
              'image_embeddings.image_distbution_embeddings.weight', 'cross_alignment.align_output.weight'):
        assert k in own, k


def test_cfg1_forward_losses_vs_golden(golden_dir):
    g = dict(np.load(os.path.join(golden_dir, 'cfg1_model.npz')))
    cfg = synth.CONFIGS['cfg1']
    m, P, sd = _build(cfg)
    m.train()
    batch = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'])
    out, mlm, rel, bce = _losses(m, batch, cfg['R'])
    assert out.shape == (cfg['R'] + cfg['T'], cfg['B'], cfg['emb_dim'])
    assert rel_l2(out.float(), g['out']) < 1e-2
    assert abs(float(mlm) - float(g['mlm_loss'])) < 5e-3
    assert abs(float(bce) - float(g['itm_bce'])) < 5e-3
    assert max_abs(rel.float(), g['rel_scores']) < 2e-2
    # padded positions are exactly zero (tensor *= mask, transformer.py:958)
    tot = batch['lengths'] + cfg['R']
    for b in range(cfg['B']):
        assert float(out[int(tot[b]):, b].abs().max()) == 0.0 if int(tot[b]) < out.shape[0] else True


@pytest.mark.parametrize('cfg_name', ['cfg1', 'mid', 'large', 'tiny', 'tiles', 'tiles768'])
def test_gradients_vs_oracle(cfg_name, monkeypatch):
    from oracle import ref_cpu as O
    cfg = _cfg(cfg_name)
    m, P, sd = _build(cfg)
    m.train()
    spy = _DispatchSpy(monkeypatch) if cfg_name.startswith('tiles') else None
    batch = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'], seed=7)
    opt_zero = m.arena().zero_grad
    opt_zero()
    out, mlm, rel, bce = _losses(m, batch, cfg['R'])
    (mlm + bce).backward()
    torch.cuda.synchronize()
    if spy is not None:
        spy.assert_benchmarked_dispatch(cfg, m)
    names = list(sd.keys())
    leaves = {n: sd[n].clone().requires_grad_(True) for n in names}
    res = O.pretrain_losses(leaves, cfg['n_layers'], cfg['n_heads'], batch, cfg['R'])
    grads = torch.autograd.grad(res['total'], [leaves[n] for n in names])
    assert rel_l2(out.float(), res['out']) < 1e-2
    assert abs(float(mlm) - float(res['mlm'])) < 5e-3 and abs(float(bce) - float(res['itm'])) < 5e-3
    own = dict(m.named_parameters())
    qb = float(dict(zip(names, grads))['attentions.0.q_lin.bias'].norm())
    bad = []
    for n, gref in zip(names, grads):
        gm = own[n].grad
        assert gm is not None, n
        if '.k_lin.bias' in n:
            assert float(gm.norm()) < 5e-2 * qb + 1e-6, n
            continue
        err = rel_l2(gm, gref)
        if err > 5e-2:
            bad.append((n, err))
    assert not bad, bad


def test_three_training_steps_track_golden(golden_dir):
    """XTrainer.pretrain_under_step x3 (clip 5, adam_inverse_sqrt): logged losses and lr follow
    the reference's own trainer run (cfg1_trainer.npz) and its 3-step model run."""
    from m3p_amd.trainer import XTrainer
    tg = dict(np.load(os.path.join(golden_dir, 'cfg1_trainer.npz')))
    cfg = synth.CONFIGS['cfg1']
    m, P, sd = _build(cfg)
    for k, v in dict(optimizer='adam_inverse_sqrt,beta1=0.9,beta2=0.98,lr=0.0001', clip_grad_norm=5, amp=-1, fp16=False,
                     accumulate_gradients=1, multi_gpu=False, epoch_size=100, cross_mlm_steps=[('google', 'img')],
                     cross_mrm_steps=[], cross_mrfr_steps=[], cross_clcm_steps=[], sample_n=2, refine_image=False,
                     multi_cls_loss_weight=0, bin_cls_loss_weight=1, batch_size=cfg['B'], dump_path='/nonexistent_m3p_dump').items():
        setattr(P, k, v)
    tr = XTrainer(m, {}, P)
    batch = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'])
    B, R = cfg['B'], cfg['R']
    img = batch['x_img'].transpose(0, 1).contiguous()
    loc = batch['image_loc'].transpose(0, 1).contiguous()
    tup = ((batch['x'], batch['lengths'], batch['x_labels']),
           (img, torch.ones(B, R, dtype=torch.long), loc, torch.full((B, R), -1), batch['pos_labels'].tolist(), None, None))
    before = {n: p.detach().clone() for n, p in m.named_parameters() if getattr(p, '_m3p_arena', None)}
    for step in range(2):
        tr.pretrain_under_step(tup, 'google', 't2i', 'en', 1.0, 1.0, 1.0, 1.0)
        assert abs(float(tr.stats['CMLM-google'][-1]) - float(tg['cmlm_step%d' % step])) < 5e-3
        assert abs(float(tr.stats['t2i-google'][-1]) - float(tg['t2i_step%d' % step])) < 5e-3
        assert abs(tr.optimizers['model'].param_groups[0]['lr'] - float(tg['lr_after%d' % step])) < 1e-15
        tr.iter()
    assert tr.stats['processed_s'] == 2 * B and tr.n_sentences == 2 * B
    # first Adam step moves every touched weight by ~lr * sign(g): check magnitude and that grads were zeroed
    after = dict(m.named_parameters())
    moved = float((after['ffns.0.lin1.weight'] - before['ffns.0.lin1.weight']).abs().mean())
    assert 0.5e-7 < moved < 5e-7, moved
    assert float(m.arena().grad.abs().max()) == 0.0
    # bf16 working copies follow the master weights
    assert torch.equal(m.arena().w('ffns.0.lin1.weight'), after['ffns.0.lin1.weight'].to(torch.bfloat16))


@pytest.mark.parametrize('cfg_name', ['cfg1', 'mid', 'tiles', 'tiles768'])
def test_dropout_on_training_step_vs_oracle_fed_the_same_masks(cfg_name, monkeypatch):
    """The benchmarked configuration (dropout = attention_dropout = 0.1) end to end: every dropout site of the
    encoder (image rows, embedding, attention probabilities, attention output, FFN output, per layer) draws its keep
    mask from the counter-based hash keyed by functional._site(); the oracle is handed those very masks (NumPy twin of
    the device RNG), so output, losses and every gradient are compared element-wise with dropout switched on -
    same bars as the dropout-free parity tests."""
    from oracle import ref_cpu as O
    cfg = _cfg(cfg_name)
    p = 0.1
    m, P, sd = _build(cfg, dropout=p)
    m.train()
    spy = _DispatchSpy(monkeypatch) if cfg_name.startswith('tiles') else None
    batch = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'], seed=7)
    m.arena().zero_grad()
    out, mlm, rel, bce = _losses(m, batch, cfg['R'])
    (mlm + bce).backward()
    torch.cuda.synchronize()
    if spy is not None:
        spy.assert_benchmarked_dispatch(cfg, m)
    keeps = encoder_keep_masks(m, m._fwd_counter, cfg['B'], cfg['T'], cfg['R'], p, p)
    frac = float(keeps['emb'].float().mean())
    assert abs(frac - (1 - p)) < 2e-2
    names = list(sd.keys())
    leaves = {n: sd[n].clone().requires_grad_(True) for n in names}
    res = O.pretrain_losses(leaves, cfg['n_layers'], cfg['n_heads'], batch, cfg['R'], dropout=p, attention_dropout=p, keeps=keeps)
    grads = dict(zip(names, torch.autograd.grad(res['total'], [leaves[n] for n in names])))
    assert rel_l2(out.float(), res['out']) < 1e-2
    assert abs(float(mlm) - float(res['mlm'])) < 5e-3 and abs(float(bce) - float(res['itm'])) < 5e-3
    # and it is not the dropout-free result
    res0 = O.pretrain_losses(sd, cfg['n_layers'], cfg['n_heads'], batch, cfg['R'])
    assert rel_l2(out.float(), res0['out']) > 5e-2
    own = dict(m.named_parameters())
    qb = float(grads['attentions.0.q_lin.bias'].norm())
    bad = []
    for n in names:
        gm = own[n].grad
        if '.k_lin.bias' in n:
            assert float(gm.norm()) < 5e-2 * qb + 1e-6, n
            continue
        err = rel_l2(gm, grads[n])
        if err > 5e-2:
            bad.append((n, err))
    assert not bad, bad


def test_dropout_training_step_runs_and_is_reproducible():
    cfg = synth.CONFIGS['cfg1']
    batch = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'])
    vals = []
    for _ in range(2):
        m, P, sd = _build(cfg, dropout=0.1)
        m.train()
        out, mlm, rel, bce = _losses(m, batch, cfg['R'])
        (mlm + bce).backward()
        vals.append((float(mlm), float(m.arena().grad.norm())))
    assert vals[0] == vals[1] or (abs(vals[0][0] - vals[1][0]) < 1e-6 and abs(vals[0][1] - vals[1][1]) < 1e-3 * vals[0][1])
    m.eval()
    o1, *_ = _losses(m, batch, cfg['R'])
    o2, *_ = _losses(m, batch, cfg['R'])
    assert torch.equal(o1, o2)


def test_gradient_accumulation_equals_the_full_batch_step():
    """accumulate_gradients = 2 (xtrainer.py:231-243: backward on every micro-step, clip + step + zero_grad on the
    boundary): two half-batch micro-steps move the weights like one step on the whole batch - the accumulated
    gradient is the sum, i.e. twice the full-batch mean, which the clip coefficient and Adam's normalisation absorb."""
    from m3p_amd.trainer import XTrainer
    cfg = dict(emb_dim=256, n_heads=4, n_layers=2, n_words=2000, T=16, R=12, B=8, n_pred=3)

    def trainer(accumulate):
        m, P, sd = _build(cfg)
        for k, v in dict(optimizer='adam_inverse_sqrt,beta1=0.9,beta2=0.98,lr=0.0001', clip_grad_norm=5, amp=1, fp16=True,
                         accumulate_gradients=accumulate, multi_gpu=False, epoch_size=100, cross_mlm_steps=[('google', 'img')],
                         cross_mrm_steps=[], cross_mrfr_steps=[], cross_clcm_steps=[], sample_n=2, refine_image=False,
                         multi_cls_loss_weight=0, bin_cls_loss_weight=1, batch_size=cfg['B'], dump_path='/nonexistent_m3p_dump').items():
            setattr(P, k, v)
        return XTrainer(m, {}, P), m

    def tup_of(batch, sl):
        n = sl.stop - sl.start
        img = batch['x_img'][:, sl].transpose(0, 1).contiguous()
        loc = batch['image_loc'][:, sl].transpose(0, 1).contiguous()
        return ((batch['x'][:, sl].contiguous(), batch['lengths'][sl], batch['x_labels'][:, sl].contiguous()),
                (img, torch.ones(n, cfg['R'], dtype=torch.long), loc, torch.full((n, cfg['R']), -1),
                 batch['pos_labels'][sl.start // 2:sl.stop // 2].tolist(), None, None))

    B = cfg['B']
    ha = synth.make_batch(cfg['T'], cfg['R'], B // 2, cfg['n_words'], cfg['n_pred'], seed=41, ragged=True)
    hb = synth.make_batch(cfg['T'], cfg['R'], B // 2, cfg['n_words'], cfg['n_pred'], seed=42, ragged=True)
    batch = {k: torch.cat([ha[k], hb[k]], dim=1) for k in ('x', 'x_labels', 'x_img', 'image_loc')}
    batch.update({k: torch.cat([ha[k], hb[k]]) for k in ('lengths', 'pos_labels')})
    names = ['attentions.0.q_lin.weight', 'ffns.1.lin2.weight', 'layer_norm_emb.weight', 'pooled_layer.dense.weight',
             'image_embeddings.image_embeddings.weight', 'position_embeddings.weight']

    tr1, m1 = trainer(1)
    before = {n: dict(m1.named_parameters())[n].detach().clone() for n in names}
    tr1.pretrain_under_step(tup_of(batch, slice(0, B)), 'google', 't2i', 'en', 1.0, 1.0, 1.0, 1.0)
    d1 = {n: dict(m1.named_parameters())[n].detach() - before[n] for n in names}

    tr2, m2 = trainer(2)
    tr2.n_iter = 1                                   # not a boundary: gradients only
    tr2.pretrain_under_step(tup_of(batch, slice(0, B // 2)), 'google', 't2i', 'en', 1.0, 1.0, 1.0, 1.0)
    mid = {n: dict(m2.named_parameters())[n].detach().clone() for n in names}
    assert all(torch.equal(mid[n], before[n]) for n in names)          # nothing moved yet
    assert float(m2.arena().grad.abs().max()) > 0.0
    tr2.n_iter = 2                                   # boundary: clip, step, zero_grad
    tr2.pretrain_under_step(tup_of(batch, slice(B // 2, B)), 'google', 't2i', 'en', 1.0, 1.0, 1.0, 1.0)
    torch.cuda.synchronize()
    assert float(m2.arena().grad.abs().max()) == 0.0
    for n in names:
        d2 = dict(m2.named_parameters())[n].detach() - before[n]
        touched = d1[n].abs() > 0
        assert rel_l2(d2[touched], d1[n][touched]) < 5e-2, n


@pytest.mark.gpu
def test_jointfwd_text_embed_override_and_its_gradient():
    """jointfwd(text_embed=...) (transformer.py:910-913): caller-made word rows instead of Emb[x].  Handing in Emb[x] itself
    must reproduce the plain pass, and the gradient that comes back for the rows must be the one the oracle's autograd gives
    (the FreeLB steps ascend along it: tests/test_streams_and_retrieval.py::test_freelb_t2i_step_tracks_the_reference_run)."""
    from oracle import ref_cpu as O
    cfg = synth.CONFIGS['cfg1']
    m, P, sd = _build(cfg)
    m.eval()
    batch = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'])
    dev = 'cuda'
    kw = dict(x=batch['x'].to(dev), lengths=batch['lengths'].to(dev), x_img=batch['x_img'].to(dev), lengths_img=batch['lengths_img'].to(dev),
              causal=False, langs=None, image_loc=batch['image_loc'].to(dev), refine_image=False)
    plain = m('jointfwd', **kw)
    rows = sd['embeddings.weight'][batch['x'].t()].to(dev).requires_grad_(True)                  # (B, T, d)
    out = m('jointfwd', text_embed=rows, **kw)
    assert rel_l2(out.float(), plain.float()) < 2e-3
    w = torch.from_numpy(np.random.RandomState(5).standard_normal(tuple(out.shape)).astype(np.float32))
    m.arena().zero_grad()
    (out.float() * w.to(dev)).sum().backward()
    leaves = {k: v.clone() for k, v in sd.items()}
    rows_ref = sd['embeddings.weight'][batch['x'].t()].clone().requires_grad_(True)
    o = O.jointfwd(leaves, cfg['n_layers'], cfg['n_heads'], batch['x'], batch['lengths'], batch['x_img'], batch['lengths_img'],
                   batch['image_loc'], text_embed=rows_ref)
    (o * w).sum().backward()
    assert rel_l2(rows.grad.float(), rows_ref.grad) < 5e-2


def test_lazy_vocab_zero_changes_nothing(monkeypatch):
    """Round 5: when a step's MLM head STORED the tied matrix's weight gradient, the fused Adam pass leaves that range of the
    gradient arena un-zeroed (the next step's store overwrites it) - `Arena.defer_vocab_zero`.  Four steps at a size where the
    store path runs - MLM + ITM, MLM + ITM, an ITM-ONLY step (no MLM head: the embedding scatter is the first writer and must
    find zeros), MLM + ITM again - end in the same weights and moments (to the run-to-run noise of the step's fp32 atomics) with
    the switch on and off, and the switch was actually exercised (a deferred range existed, was cleared by a store once and by a memset once)."""
    from m3p_amd import functional as Fn, optim as Om
    cfg = dict(emb_dim=768, n_heads=12, n_layers=1, n_words=88000, T=96, R=32, B=128, n_pred=32)     # n = 4096 (every ragged length >= 48 holds 32 targets), 1032 output tiles
    batch = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'], seed=3)
    assert batch['y'].numel() == 4096
    results, events = [], []
    for lazy in (True, False, False):      # (the third run measures the step's own run-to-run noise)
        monkeypatch.setattr(Om, '_LAZY_VOCAB_ZERO', lazy)
        m, P, sd = _build(cfg, dropout=0.1)
        m.train()
        opt = Om.get_optimizer(m.parameters(), 'adam_inverse_sqrt,beta1=0.9,beta2=0.98,lr=0.0001')
        ar = m.arena()
        ev = []
        real_zero, real_defer = ar.ensure_zero, ar.defer_vocab_zero

        def ensure_zero():
            if ar.stale is not None:
                ev.append('memset')
            real_zero()

        def defer():
            ev.append('defer')
            real_defer()
        monkeypatch.setattr(ar, 'ensure_zero', ensure_zero)
        monkeypatch.setattr(ar, 'defer_vocab_zero', defer)
        for step in range(4):
            out, mlm, rel, bce = _losses(m, batch, cfg['R'])
            (bce if step == 2 else mlm + bce).backward()
            opt.clip_grad_norm(5.0)
            opt.step()
        torch.cuda.synchronize()
        ar.ensure_zero()
        assert float(ar.grad.abs().max()) == 0.0
        results.append((ar.master.clone(), [e['m'].clone() for e in opt._arenas.values()], [e['v'].clone() for e in opt._arenas.values()]))
        events.append(ev)
    assert events[0].count('defer') == 3 and events[0].count('memset') == 2 and events[1] == [], events
    # (not bit-for-bit: the step's fp32 atomics - embedding scatter, LayerNorm / bias column sums - add in a different order from
    #  run to run; a stale vocabulary range would show up at the size of a whole gradient, orders of magnitude above these bars)
    assert events[2] == []
    for k, name in ((0, 'master'), (1, 'exp_avg'), (2, 'exp_avg_sq')):
        pick = (lambda r: [r[0]]) if k == 0 else (lambda r, k=k: r[k])      # noqa: E731
        for a, b, c in zip(pick(results[0]), pick(results[1]), pick(results[2])):
            noise, diff = rel_l2(c, b), rel_l2(a, b)
            assert diff <= max(10 * noise, 1e-7), (name, diff, noise)
