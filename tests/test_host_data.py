"""Data formats on the input side of the text-only and generation steps against goldens recorded from the reference
(tests/golden/host_data.npz, oracle/gen_goldens.py::gen_data_goldens): the token-stream dataset behind ``mlm_step``
(lane matrix, plain / shuffled / resumed epochs, sub-selection, language ids) and the captioning, multimodal-translation,
text-to-text and sliding-window collates.  Integer work: everything is compared bit for bit."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from m3p_amd import collate, synth
from m3p_amd.datasets import StreamDataset


@pytest.fixture(scope='module')
def G(golden_dir):
    return np.load(os.path.join(golden_dir, 'host_data.npz'), allow_pickle=False)


def _params():
    return SimpleNamespace(bptt=8, batch_size=5, eos_index=synth.EOS, n_gpu_per_node=2, local_rank=1, lang2id={'en': 0, 'zh': 1})


def _epoch_matches(G, tag, it):
    n = 0
    for i, b in enumerate(it):
        assert b[0].dtype == torch.int64 and np.array_equal(b[0].numpy(), G['%s.%d.x' % (tag, i)]), (tag, i)
        assert np.array_equal(b[1].numpy(), G['%s.%d.len' % (tag, i)])
        if len(b) > 2:
            assert np.array_equal(b[2].numpy(), G['%s.%d.langs' % (tag, i)])
        n += 1
    assert n == int(G[tag + '.n'])


def test_token_stream_is_the_one_the_golden_was_made_from(G):
    sent, pos, langs = synth.token_stream(seed=41)
    assert np.array_equal(sent, G['sd_sent']) and np.array_equal(pos, G['sd_pos']) and np.array_equal(langs, G['sd_langs'])


def test_stream_dataset_epochs_match_the_reference(G):
    sent, pos, _ = synth.token_stream(seed=41)
    ds = StreamDataset(sent, pos, _params())
    assert ds.data.dtype == G['sd_data'].dtype and np.array_equal(ds.data, G['sd_data'])
    assert [ds.n_tokens, ds.n_batches, len(ds)] == G['sd_counts'].tolist()
    # lane layout: lane b is a contiguous slice of the EOS-left-padded stream, one EOS row in front
    rows = ds.n_batches * ds.bptt
    flat = np.concatenate([np.full(rows * 5 - len(sent), synth.EOS, sent.dtype), sent])
    assert (ds.data[0] == synth.EOS).all() and np.array_equal(ds.data[1:, 3], flat[3 * rows:4 * rows])
    _epoch_matches(G, 'sd_plain', ds.get_iterator(shuffle=False))
    _epoch_matches(G, 'sd_shuf1', ds.get_iterator(shuffle=True, seed=17))
    _epoch_matches(G, 'sd_shuf2', ds.get_iterator(shuffle=True, seed=17))
    assert ds.loaded[1] == G['sd_loaded'].tolist() and ds.loaded[0] == []
    # a reloaded run re-walks the permutation of the interrupted epoch and skips what that epoch had served
    ds.reload_check({0: [], 1: [int(ds.n_batches), 2]})
    _epoch_matches(G, 'sd_resume', ds.get_iterator(shuffle=True, seed=17))
    assert ds.loaded[1] == G['sd_resume_loaded'].tolist() and ds.reload is False
    assert int(G['sd_resume.n']) == ds.n_batches - 2
    ds.select_data(1, 3)
    assert np.array_equal(ds.data, G['sd_sel_data']) and [ds.n_batches, len(ds)] == G['sd_sel_counts'].tolist()
    n = ds.n_batches
    ds.select_data(2, 9)                                # invalid split: warned about, nothing changes
    assert ds.n_batches == n


def test_stream_dataset_language_ids_and_subsampling(G):
    sent, pos, langs = synth.token_stream(seed=41)
    ds = StreamDataset(sent, pos, _params(), langs=langs)
    assert ds.has_lan and np.array_equal(ds.langs, G['sd_lang_matrix'])
    assert ds.langs.shape[0] == ds.data.shape[0] - 1    # the language matrix has no leading row (dataset_pretrain.py:819-821)
    _epoch_matches(G, 'sd_lang', ds.get_iterator(shuffle=False, subsample=2))
    assert int(G['sd_lang.n']) == ds.n_batches // 2


def test_stream_dataset_checks_its_input():
    sent, pos, _ = synth.token_stream(seed=41)
    with pytest.raises(AssertionError):
        StreamDataset(sent, pos[:-1], _params())        # a sentence without its position pair
    bad = sent.copy()
    bad[pos[2, 1]] = 77                                  # a position that does not end on EOS
    with pytest.raises(AssertionError):
        StreamDataset(bad, pos, _params())
    # a stream shorter than one batch still yields one (EOS-padded) batch
    ds = StreamDataset(sent[:pos[1, 1] + 1], pos[:2], _params())
    (x, lengths), = list(ds.get_iterator(shuffle=False))
    assert x.shape == (8, 5) and lengths.tolist() == [8] * 5 and int((x != synth.EOS).sum()) <= pos[1, 1] + 1


def test_mlm_step_reads_the_stream_dataset():
    """generate_batch('pred') -> (x, lengths) of the stream iterator, language ids filled in for multilingual runs
    (xtrainer.py:485-509): the host side of mlm_step up to the masked batch, no GPU involved."""
    from m3p_amd import masking
    from m3p_amd.trainer import Trainer
    sent, pos, _ = synth.token_stream(seed=41)
    P = _params()
    ds = StreamDataset(sent, pos, P)
    tr = Trainer.__new__(Trainer)
    tr.params = SimpleNamespace(langs=['en', 'zh'], lang2id={'en': 0, 'zh': 1}, n_langs=2, group_by_size=False)
    tr.data = {'mono_stream': {'zh': {'train': ds}}}
    tr.iterators = {}
    np.random.seed(3)
    seen = []
    for _ in range(ds.n_batches + 1):                    # one more than an epoch: the iterator is re-created
        x, lengths, positions, langs, _ = tr.generate_batch('zh', None, 'pred')
        assert x.shape == (8, 5) and positions is None and (langs == 1).all() and lengths.tolist() == [8] * 5
        seen.append(x)
    assert len(ds.loaded[1]) == 2 and ds.loaded[1][0] == ds.n_batches
    M = SimpleNamespace(sample_alpha=0, fp16=False, mask_scores=None, word_pred=0.15, pad_index=synth.PAD, n_words=1000,
                        mask_index=999, pred_probs=torch.FloatTensor([0.8, 0.1, 0.1]))
    np.random.seed(5); torch.manual_seed(5)
    x2, y, pm = masking.mask_out(seen[0].clone(), lengths, M)
    assert int(pm.sum()) == len(y) > 0 and torch.equal(seen[0][pm], y)


# ---- generation collates ------------------------------------------------------------------------------------------
def _tree(G, prefix):
    keys = [k for k in G.files if k == prefix or k.startswith(prefix + '.')]
    if keys == [prefix]:
        return G[prefix]
    kids = sorted({int(k[len(prefix) + 1:].split('.')[0]) for k in keys})
    return [_tree(G, '%s.%d' % (prefix, i)) if i in kids else [] for i in range(max(kids) + 1)]


def _same(a, b, path='root'):
    if isinstance(b, list):
        assert isinstance(a, (list, tuple)) and len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, '%s.%d' % (path, i))
    else:
        got = a.numpy() if torch.is_tensor(a) else np.asarray(a)
        assert got.shape == b.shape and np.array_equal(got, b), path


def _words(v):
    return np.asarray(v, dtype=np.int64).reshape(-1)


def test_generation_collates_match_the_reference(G):
    t = torch.from_numpy
    cap = [(_words(f[0]), t(f[1]), t(f[2]), t(f[3]), int(f[4])) for f in _tree(G, 'col_cap_in')]
    _same(collate.caption_collate(cap), _tree(G, 'col_cap'))
    (x, lengths), (img, mask, loc, ids) = collate.caption_collate(cap)
    assert x.shape[1] == 4 and img.shape == (4, 4, 2048) and mask.shape == (4, 4) and loc.shape == (4, 4, 5) and len(ids) == 4
    assert (x[0] == synth.BOS).all() and all(int(x[int(n) - 1, i]) == synth.EOS for i, n in enumerate(lengths))

    mtc = [(_words(f[0]), _words(f[1]), t(f[2]), t(f[3]), t(f[4]), int(f[5])) for f in _tree(G, 'col_mtc_in')]
    _same(collate.mt_caption_collate(mtc), _tree(G, 'col_mtc'))

    ntg = [(_words(f[0]), _words(f[1])) for f in _tree(G, 'col_ntg_in')]
    _same(collate.ntg_collate(ntg), _tree(G, 'col_ntg'))

    sld = [([_words(c) for c in f[0]], t(f[1]), t(f[2]), t(f[3]), [int(v) for v in f[4]], [int(v) for v in f[5]])
           for f in _tree(G, 'col_sld_in')]
    _same(collate.slide_collate(sld), _tree(G, 'col_sld'))
    (x, lengths), (img, mask, loc, ids), labels = collate.slide_collate(sld)
    assert x.shape[1] == 6 == img.shape[0] == len(ids) == len(labels)


def test_loader_picks_the_collate_the_flags_name():
    """xtrainer.py:1164-1181: is_generation (+ is_mt) / is_pretrain / is_slide choose the collate of the cross-modal loader."""
    from m3p_amd.trainer import XTrainer
    rs = np.random.RandomState(0)

    class Items(torch.utils.data.Dataset):
        def __init__(self, make):
            self.items = [make() for _ in range(5)]

        def __len__(self):
            return len(self.items)

        def __getitem__(self, i):
            return self.items[i]

    def regions(k=1, R=3):
        return (torch.from_numpy(rs.standard_normal((k, R, 2048)).astype(np.float32)), torch.ones(k, R, dtype=torch.long),
                torch.zeros(k, R, 5))

    def words():
        return rs.randint(4, 900, size=rs.randint(1, 6)).astype(np.int64)

    tr = XTrainer.__new__(XTrainer)
    tr.epoch, tr.iterators = 0, {}
    tr.params = SimpleNamespace(batch_size=2, n_gpu_per_node=1, num_workers=0, is_generation=True, is_mt=False, is_pretrain=False,
                                is_slide=False)
    tr.data = {'cross_modal': {('coco', 'img'): {'train': Items(lambda: (words(),) + regions() + (7,))}}}
    (x2, len2), (x1, x1_mask, loc, ids) = tr.get_batch('txt2img', 'coco', 'img')
    assert x2.shape[1] == 2 and x1.shape == (2, 3, 2048) and x1_mask.shape == (2, 3) and len(ids) == 2
    tr.params.is_mt, tr.iterators = True, {}
    tr.data = {'cross_modal': {('coco', 'img'): {'train': Items(lambda: (words(), words()) + regions() + (7,))}}}
    (xs, ls), (x2, len2), (x1, x1_mask, loc, ids) = tr.get_batch('txt2img', 'coco', 'img')
    assert xs.shape[1] == x2.shape[1] == 2 and int(ls.max()) == xs.shape[0]
    # five items in batches of two: the third batch is the short one, the fourth call starts a new epoch
    sizes = [tr.get_batch('txt2img', 'coco', 'img')[0][0].shape[1] for _ in range(3)]
    assert sorted(sizes) == [1, 2, 2]


def test_span_masking_batches_match_the_reference_bit_for_bit(golden_dir):
    """restricted_mask_sent (MASS) and bart_token_mask_sent (text infilling), xtrainer.py:1207-1381, under the numpy / random /
    torch seeds the golden was recorded with (tests/golden/host_spans.npz): seven tensors per call, three draws per case."""
    import random
    from m3p_amd import masking
    G = np.load(os.path.join(golden_dir, 'host_spans.npz'))
    P = SimpleNamespace(word_mass=0.5, pad_index=synth.PAD, mask_index=999, n_words=1000, pred_probs=torch.FloatTensor([0.8, 0.1, 0.1]))
    n = 0
    for case in range(4):
        x, lengths, min_len = torch.from_numpy(G['%d.x' % case]), torch.from_numpy(G['%d.len' % case]), int(G['%d.min_len' % case])
        for name, fn in (('mass', masking.restricted_mask_sent), ('bart', masking.bart_token_mask_sent)):
            for rep in range(3):
                seed = 500 + 10 * case + rep
                np.random.seed(seed); random.seed(seed); torch.manual_seed(seed)
                res = fn(x.clone(), lengths.clone(), P, min_len if name == 'mass' else 100000)
                for tag, v in zip(('x1', 'len1', 'x2', 'len2', 'y', 'pred_mask', 'pos'), res):
                    want = G['%d.%s.%d.%s' % (case, name, rep, tag)]
                    assert v.numpy().shape == want.shape and np.array_equal(v.numpy(), want), (case, name, rep, tag)
                    n += 1
                x1, len1, x2, len2, y, pred_mask, pos = res
                if name == 'bart':          # one <mask> per sentence, the decoder sees the whole original sentence
                    assert ((x1 == P.mask_index).sum(0) == 1).all() and torch.equal(len2, lengths - 1)
                    assert all(torch.equal(x2[:int(len2[b]), b], x[:int(len2[b]), b]) for b in range(x.size(1)))
                    assert int(pred_mask.sum()) == int(len2.sum()) == len(y)
                else:                       # the same number of words masked in every sentence, never the first symbol
                    assert len(set(len2.tolist())) == 1 and torch.equal(len1, lengths) and (x1[0] == x[0]).all()
                    assert torch.equal(y, torch.cat([x[pos[:, b] + 1, b] for b in range(x.size(1))]).view(x.size(1), -1).t()[pred_mask])
    assert n == 4 * 2 * 3 * 7


def test_region_feature_noise_matches_the_reference(golden_dir):
    """bart_img_noise / _mask_object (xtrainer.py:1699-1744) under the reference's numpy / random seeds: how many regions stay,
    which are blanked, the (re-normalised) features, boxes and mask (tests/golden/host_img_noise.npz)."""
    import random
    from m3p_amd import masking
    G = np.load(os.path.join(golden_dir, 'host_img_noise.npz'))
    for case in range(3):
        B, R = G['%d.shape' % case].tolist()
        rs = np.random.RandomState(53)
        for c in range(case + 1):               # the generator drew the cases one after another from one stream
            b_, r_ = [(4, 10), (3, 36), (2, 7)][c]
            feats = torch.from_numpy(rs.standard_normal((b_, r_, 2048)).astype(np.float32))
            feats = feats / feats.norm(dim=-1, keepdim=True)
            loc = torch.from_numpy(rs.uniform(size=(b_, r_, 5)).astype(np.float32))
        mask = torch.ones(B, R, dtype=torch.long)
        for rep in range(3):
            seed = 700 + 10 * case + rep
            np.random.seed(seed); random.seed(seed)
            f2, l2, m2 = masking.bart_img_noise(feats.clone(), loc.clone(), mask.clone())
            assert f2.shape == (B, int(G['%d.%d.n' % (case, rep)]), 2048) and f2.shape[1] < R
            assert np.array_equal((f2.abs().sum(-1) == 0).numpy(), G['%d.%d.blank' % (case, rep)])
            assert np.array_equal(f2[:, :, :8].numpy(), G['%d.%d.first8' % (case, rep)])
            assert np.allclose(f2.double().sum(-1).numpy(), G['%d.%d.sum' % (case, rep)], rtol=0, atol=1e-12)
            assert np.array_equal(l2.numpy(), G['%d.%d.loc' % (case, rep)]) and np.array_equal(m2.numpy(), G['%d.%d.mask' % (case, rep)])
            norms = f2.norm(dim=-1)
            assert (((norms - 1).abs() < 1e-5) | (norms == 0)).all()
