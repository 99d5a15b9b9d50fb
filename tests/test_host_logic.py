"""Host-side logic that needs no GPU: optimizer spec parsing / lr schedule, collate,
mask building, the dropout RNG twin, arena layout bookkeeping."""
import numpy as np
import pytest
import torch

from m3p_amd import rng, synth


def test_get_optimizer_spec_and_schedule():
    from m3p_amd.optim import get_optimizer, AdamInverseSqrtWithWarmup
    p = [torch.nn.Parameter(torch.zeros(4))]
    opt = get_optimizer(p, 'adam_inverse_sqrt,beta1=0.9,beta2=0.98,lr=0.0001')
    assert isinstance(opt, AdamInverseSqrtWithWarmup)
    g = opt.param_groups[0]
    assert g['betas'] == (0.9, 0.98) and g['lr'] == 1e-7 and g['num_updates'] == 0
    assert abs(opt.get_lr_for_step(1) - 1.24975e-07) < 1e-15
    assert abs(opt.get_lr_for_step(4000) - 1e-4) < 1e-12
    assert abs(opt.get_lr_for_step(16000) - 0.5e-4) < 1e-12
    with pytest.raises(Exception):
        get_optimizer(p, 'adam,foo=1')
    with pytest.raises(Exception):
        get_optimizer(p, 'nosuch,lr=1')
    # eager state like the reference (optim.py:35-40)
    assert opt.state[p[0]]['step'] == 0 and opt.state[p[0]]['exp_avg'].shape == (4,)


def test_cpu_fallback_param_path_matches_oracle():
    """Parameters outside any arena take the reference loop (kept for completeness)."""
    from m3p_amd.optim import get_optimizer
    from oracle import ref_cpu as O
    w = torch.nn.Parameter(torch.tensor([0.5, -1.0, 2.0]))
    opt = get_optimizer([w], 'adam,lr=0.01,beta1=0.9,beta2=0.98,weight_decay=0.01')
    p, m, v = w.detach().clone(), torch.zeros(3), torch.zeros(3)
    for step in range(1, 4):
        g = torch.tensor([0.1 * step, -0.2, 0.3])
        w.grad = g.clone()
        opt.step()
        p, m, v = O.adam_step(p, g, m, v, step, 0.01, 0.9, 0.98, 1e-8, 0.01)
        assert torch.allclose(w.detach(), p, rtol=1e-6, atol=1e-8)


def test_batch_sentences_v2_and_get_mask():
    from m3p_amd.trainer import batch_sentences_v2
    sents = [np.array([5, 6, 7]), np.array([9]), np.array([], dtype=np.int64)]
    labels = [[-1, 6, -1], [9], []]
    x, lens, lab = batch_sentences_v2(sents, labels)
    assert lens.tolist() == [5, 3, 2] and x.shape == (5, 3)
    assert x[:, 0].tolist() == [0, 5, 6, 7, 2] and x[:, 1].tolist() == [0, 9, 2, 1, 1] and x[:, 2].tolist() == [0, 2, 1, 1, 1]
    assert lab[:, 0].tolist() == [-1, -1, 6, -1, -1]
    pm = lab != -1
    y = lab[lab > 0]
    assert int(pm.sum()) == 2 and y.tolist() == [9, 6] or y.tolist() == [6, 9]


def test_rng_twin_statistics_and_determinism():
    k1 = rng.keep_mask(200000, 123, 0.1)
    k2 = rng.keep_mask(200000, 123, 0.1)
    assert np.array_equal(k1, k2)
    assert abs(k1.mean() - 0.9) < 3e-3
    k3 = rng.keep_mask(200000, 124, 0.1)
    assert abs((k1 == k3).mean() - (0.81 + 0.01)) < 5e-3          # independent streams
    assert rng.keep_mask(1000, 5, 0.0).all()
    assert rng.stream_seed(1, 2, 3) != rng.stream_seed(1, 2, 4) != rng.stream_seed(1, 3, 3)
    # known-answer for the hash itself, restated in plain Python integers (csrc/common.hpp: m3p_hash32; the GPU tests pin the C
    # version to this twin through every dropout site's mask)
    def ref(idx, seed):
        h = (idx + seed) & 0xFFFFFFFF
        for k in (0x9E3779, 0x85EBCB, 0xC2B2AF):
            h ^= h >> 16
            h = (h + (h & 0xFFFFFF) * k) & 0xFFFFFFFF
        return h ^ (h >> 16)
    assert int(rng.hash32(np.array([0], dtype=np.uint64), 0)[0]) == 0
    for idx, seed in ((1, 7), (12345, 0), (0x3FFFFFF, 0xDEADBEEF), (41_000_000, 0xFFFFFFFF)):
        assert int(rng.hash32(np.array([idx], dtype=np.uint64), seed)[0]) == ref(idx, seed)
    assert ref(1, 7) == 0xB9D534D5      # (one literal, so that twin and restatement cannot drift together)


def test_synthetic_batch_contract():
    cfg = synth.CONFIGS['cfg1']
    b = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'])
    assert b['x'].shape == (cfg['T'], cfg['B']) and b['x'].dtype == torch.int64
    assert int(b['lengths'].max()) == cfg['T'] and (b['x'][0] == 0).all()
    assert int(b['pred_mask'].sum()) == b['y'].numel() == cfg['n_pred'] * cfg['B']
    assert (b['y'] >= 4).all() and (b['x'][b['pred_mask']] == cfg['n_words'] - 1).all()
    assert torch.allclose(b['x_img'].norm(dim=-1), torch.ones(cfg['R'], cfg['B']), atol=1e-5)
    for i in range(cfg['B']):
        n = int(b['lengths'][i])
        assert b['x'][n - 1, i] == 2 and (b['x'][n:, i] == 1).all()


def test_model_constructs_on_cpu_with_reference_names():
    """Constructor + state-dict surface work without a GPU; running the hot path does not."""
    from m3p_amd.model.transformer import TransformerModel
    P = synth.model_params(64, 2, 2, 120, refine_layers=1)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True)
    sd = m.state_dict()
    for k, shape in synth.hot_param_shapes(P).items():
        assert tuple(sd[k].shape) == tuple(shape), k
    assert 'refine_embeddings.layers.0.self_attn.aoa_layer.0.weight' in sd
    assert m.pred_layer.proj.weight is m.embeddings.weight
    assert float(m.embeddings.weight[P.pad_index].abs().sum()) == 0.0
    with pytest.raises(RuntimeError):
        m('jointfwd', x=torch.zeros(4, 2, dtype=torch.long), lengths=torch.tensor([4, 4]),
          x_img=torch.zeros(2, 2, 2048), lengths_img=torch.tensor([2, 2]), image_loc=torch.zeros(2, 2, 5))
    with pytest.raises(NotImplementedError):
        m('fwd', x=None)


def test_state_dict_is_the_reference_enumeration_and_checkpoints_round_trip(golden_dir, tmp_path):
    """Checkpoint interop: our state_dict() has exactly the reference model's keys and shapes
    (tests/golden/state_dict_enum.npz, recorded from the reference), so a released checkpoint's 'model' entry loads
    strictly; 'module.'-prefixed keys (saved from under DDP, model/__init__.py:99-100) and the reference's
    {'model': ..., 'params': ...} layout (xtrainer.py:517-529) go through save_model / reload_checkpoint."""
    import os
    import numpy as np
    from types import SimpleNamespace
    from m3p_amd.model.transformer import TransformerModel
    from m3p_amd.trainer import XTrainer
    g = np.load(os.path.join(golden_dir, 'state_dict_enum.npz'))
    d, h, nl, V, nref = (int(v) for v in g['geometry'])
    P = synth.model_params(d, h, nl, V, refine_layers=nref)
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True)
    sd = m.state_dict()
    ref = {str(k): tuple(int(v) for v in s if v) for k, s in zip(g['keys'], g['shapes'])}
    assert sorted(sd.keys()) == sorted(ref.keys())
    for k, shape in ref.items():
        assert tuple(sd[k].shape) == shape, k
    # a checkpoint in the reference's layout with DDP-prefixed keys
    torch.manual_seed(3)
    want = {k: torch.randn_like(v) for k, v in sd.items()}
    want['pred_layer.proj.weight'] = want['embeddings.weight']          # tied (transformer.py:728-729)
    path = os.path.join(str(tmp_path), 'checkpoint.pth')
    torch.save({'model': {'module.' + k: v for k, v in want.items()}, 'params': dict(P.__dict__), 'epoch': 4,
                'n_total_iter': 17, 'best_metrics': {}, 'best_stopping_criterion': None,
                'model_optimizer': {'param_groups': [{'num_updates': 123, 'lr': 0.0}]}}, path)
    for k, v in dict(optimizer='adam_inverse_sqrt,beta1=0.9,beta2=0.98,lr=0.0001', clip_grad_norm=5, amp=-1, fp16=False,
                     accumulate_gradients=1, multi_gpu=False, local_rank=0, epoch_size=10, batch_size=2,
                     dump_path=str(tmp_path)).items():
        setattr(P, k, v)
    tr = XTrainer(m, {}, P)          # the constructor finds dump_path/checkpoint.pth like the reference (xtrainer.py:133, :566)
    got = m.state_dict()
    for k, v in want.items():
        assert torch.equal(got[k], v), k
    assert tr.epoch == 5 and tr.n_total_iter == 17
    grp = tr.optimizers['model'].param_groups[0]
    assert grp['num_updates'] == 123 and abs(grp['lr'] - tr.optimizers['model'].get_lr_for_step(123)) < 1e-12
    out = tr.save_model('best')
    back = torch.load(out, map_location='cpu', weights_only=False)
    assert set(back.keys()) == {'model', 'params'} and sorted(back['model'].keys()) == sorted(ref.keys())
    assert all(torch.equal(back['model'][k], want[k]) for k in want)


def test_store_range_check_sees_every_parameter_the_pad_rows_reach():
    """MLMHeadFn.backward may STORE the tied matrix's weight gradient over grad[o : o + V_pad * d] only if nothing that overlaps the
    range has been written since zero_grad - the pad rows reach past the matrix into whatever follows it in the arena (ADVICE r4:
    with a wide model and few pad columns' worth of bias that is position_embeddings' gradient, not only the bias's)."""
    from types import SimpleNamespace
    from m3p_amd.functional import Arena
    V, d, V_pad = 1000, 64, 1024                       # pad rows: 24 * 64 = 1536 elements > V = 1000 bias elements
    offsets = {'embeddings.weight': (0, V * d, (V, d)), 'pred_layer.proj.bias': (V * d, V, (V,)),
               'position_embeddings.weight': (V * d + 1024, 512 * d, (512, d)), 'layer_norm_emb.weight': (V * d + 1024 + 512 * d, d, (d,))}
    ar = SimpleNamespace(offsets=offsets, touched=set(), planned=set(), grads_known_zero=True)
    fresh = lambda: Arena.range_untouched(ar, 0, V_pad * d)    # noqa: E731
    assert fresh()
    ar.touched = {'layer_norm_emb.weight'}              # behind the range: harmless
    assert fresh()
    ar.touched = {'position_embeddings.weight'}         # a no-MLM accumulation micro-step wrote it; the pad rows would wipe it
    assert not fresh()
    ar.touched = {'pred_layer.proj.bias'}
    assert not fresh()
    ar.touched = {'embeddings.weight'}
    assert not fresh()
    # round 6: a parameter that data parallelism only ANNOUNCED for this step (Arena.plan: every rank steps the heads some rank may
    # train) has not been written - the store is still allowed - until a real writer touches it
    ar.touched = set()
    Arena.plan(ar, 'embeddings.weight', 'pred_layer.proj.bias')
    assert ar.touched == {'embeddings.weight', 'pred_layer.proj.bias'} and fresh()
    Arena.touch(ar, 'pred_layer.proj.bias')
    assert not fresh() and ar.planned == {'embeddings.weight'}
    Arena.plan(ar, 'pred_layer.proj.bias')              # announcing what has been written does not make it unwritten
    assert not fresh()
