"""Host-side batch preparation of the text MLM step (``Trainer.mlm_step``): which words are
predicted and what replaces them (M3P/src/xtrainer.py:385-434 ``mask_out``), and the fp16
batch rounding to multiples of 8 (:654-692 ``round_batch``).

These are integer / RNG computations on CPU tensors; parity with the reference is BIT-EXACT and
that includes the random streams: the draws below are made in the reference's order from the
same global generators (``np.random`` for the selection, ``torch`` for the 80/10/10 choice, the
random replacement ids and the sentence subsample), so a seeded reference run and a seeded run
of this module produce identical batches (tests/golden/masking.npz)."""
import math

import numpy as np
import torch


def select_targets(x, params):
    """(slen, bs) uint8 mask of the positions to predict, before the pad / first-row / fp16 trimming:
    Bernoulli(word_pred) per position, or - with sample_alpha != 0 - exactly ceil(word_pred * slen * bs)
    positions drawn without replacement proportionally to params.mask_scores[token id]."""
    slen, bs = x.shape
    if params.sample_alpha == 0:
        return torch.from_numpy((np.random.rand(slen, bs) <= params.word_pred).astype(np.uint8))
    weight = params.mask_scores[x.flatten()]
    picked = np.random.choice(len(weight), math.ceil(params.word_pred * slen * bs), replace=False, p=weight / weight.sum())
    flat = torch.zeros(slen * bs, dtype=torch.uint8)
    flat[picked] = 1
    return flat.view(slen, bs)


def mask_out(x, lengths, params):
    """-> (x with the selected words replaced, original ids of the selected words, bool mask (slen, bs)).
    Replacement per selected word: <mask> / unchanged / a random id, drawn from params.pred_probs
    (word_mask, word_keep, word_rand).  Padding and the first row are never selected; under fp16 the number of
    targets is trimmed to a multiple of 8 by dropping the earliest ones; an empty selection predicts [0, 0]."""
    slen, bs = x.shape
    sel = select_targets(x, params)
    sel[x == params.pad_index] = 0
    sel[0] = 0
    if params.fp16:
        flat = sel.view(-1)
        n = int(flat.sum())
        keep = max(n % 8, 8 * (n // 8))
        if keep != n:
            flat[torch.nonzero(flat).view(-1)[:n - keep]] = 0
        sel = flat.view(slen, bs)
    pred_mask = sel.bool()
    real = x[pred_mask]
    if real.numel() == 0:
        pred_mask[0, 0] = 1
        real = x[pred_mask]
    rand = real.clone().random_(params.n_words)
    choice = torch.multinomial(params.pred_probs, len(real), replacement=True)
    new = torch.where(choice == 0, torch.full_like(real, params.mask_index), torch.where(choice == 1, real, rand))
    x = x.masked_scatter(pred_mask, new)
    assert 0 <= int(x.min()) and int(x.max()) < params.n_words
    return x, real, pred_mask


def round_batch(x, lengths, positions, langs, params):
    """fp16 only: subsample the sentences to a multiple of 8 (random subset) and pad the length to a multiple
    of 8.  -> (x, lengths, positions, langs, kept sentence indices or None)."""
    if not params.fp16 or len(lengths) < 8:
        return x, lengths, positions, langs, None
    n_in = len(lengths)
    n_out = 8 * (n_in // 8)
    idx = None
    if n_out != n_in:
        idx = torch.randperm(n_in)[:n_out]
        lengths = lengths[idx]
        slen = int(lengths.max())
        x = x[:slen, idx]
        positions = positions[:slen, idx] if positions is not None else None
        langs = langs[:slen, idx] if langs is not None else None
    extra = -x.size(0) % 8
    if extra:
        x = torch.cat([x, torch.full((extra, n_out), params.pad_index, dtype=torch.long)], 0)
        if positions is not None:
            positions = torch.cat([positions, torch.arange(extra)[:, None] + positions[-1][None] + 1], 0)
        if langs is not None:
            langs = torch.cat([langs, langs[-1][None].expand(extra, n_out)], 0)
    assert x.size(0) % 8 == 0 and x.size(1) % 8 == 0
    return x, lengths, positions, langs, idx


def word_shuffle(x, lengths, k):
    """Local word shuffle of the denoising auto-encoder input (xtrainer.py:291-310): word j of a sentence moves to the rank
    of j + U(0, k) among its sentence (the first symbol gets -1 and stays; the last symbol is not part of the window).
    One np.random.uniform draw of shape (slen - 1, bs), like the reference."""
    if k == 0:
        return x, lengths
    assert k > 1
    noise = np.random.uniform(0, k, size=(x.size(0) - 1, x.size(1)))
    noise[0] = -1
    out = x.clone()
    for b, n in enumerate(lengths.tolist()):
        order = (np.arange(n - 1) + noise[:n - 1, b]).argsort()
        out[:n - 1, b] = x[:n - 1, b][torch.from_numpy(order)]
    return out, lengths


def word_dropout(x, lengths, p, pad_index):
    """Random word removal (xtrainer.py:312-345): every word but the first is dropped with probability p; the window is
    the sentence WITHOUT its final symbol, which the reference does not put back (its re-append is commented out), so
    every sentence comes back at least one symbol shorter; a sentence reduced to its first symbol gets one random word
    of its own back.  RNG order: one np.random.rand of shape (slen - 1, bs), then one randint per emptied sentence."""
    if p == 0:
        return x, lengths
    assert 0 < p < 1
    keep = np.random.rand(x.size(0) - 1, x.size(1)) >= p
    keep[0] = True
    kept = []
    for b, n in enumerate(lengths.tolist()):
        words = x[:n - 1, b].tolist()
        s = [w for j, w in enumerate(words) if keep[j, b]]
        if len(s) == 1:
            s.append(words[np.random.randint(1, len(words))])
        kept.append(s)
    new_len = torch.LongTensor([len(s) for s in kept])
    out = torch.full((int(new_len.max()), len(kept)), pad_index, dtype=torch.long)
    for b, s in enumerate(kept):
        out[:len(s), b] = torch.LongTensor(s)
    return out, new_len


def add_noise(x, lengths, params):
    """xtrainer.py:376-383: shuffle, then dropout (the blanking pass is disabled in the reference)."""
    x, lengths = word_shuffle(x, lengths, getattr(params, 'word_shuffle', 0))
    return word_dropout(x, lengths, getattr(params, 'word_dropout', 0), params.pad_index)
