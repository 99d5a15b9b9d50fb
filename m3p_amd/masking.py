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


# ---- span masking of the sequence-to-sequence denoising steps (xtrainer.py:1207-1381) ---------------------------------------
# A sentence of n symbols keeps its first one; `mask_len` of the others are chosen as whole spans of at most `min_len`
# words.  RNG order per sentence, as in the reference: one np.random.random() for the span placement rule, one
# random.shuffle (Python's generator) of the span / gap list, and - MASS only - np.random.randint + torch.multinomial for
# the 80 / 10 / 10 replacement of the chosen words.

def span_lengths(mask_len, min_len):
    """``get_segments`` (:1260-1267): mask_len split into spans of min_len, the remainder last."""
    spans = [min_len] * (mask_len // min_len)
    if mask_len % min_len:
        spans.append(mask_len % min_len)
    return spans


def place_spans(spans, gaps):
    """``shuffle_segments`` (:1237-1258): the spans and the single-word gaps (zeros) in random order; with probability
    0.2 the first span is pinned to the sentence start, with 0.2 the last span to its end."""
    import random
    p = np.random.random()
    if p >= 0.8:
        body = spans[1:] + gaps
    elif p >= 0.6:
        body = spans[:-1] + gaps
    else:
        body = spans + gaps
    random.shuffle(body)
    if p >= 0.8:
        return spans[0:1] + body
    if p >= 0.6:
        return body + spans[-1:]
    return body


def span_positions(layout):
    """``unfold_segments`` (:1217-1235): positions (from 1: the first symbol is never masked) covered by the spans of a
    layout in which an entry l >= 1 is a span of l words and 0 a single unmasked word."""
    pos, cur = [], 1
    for l in layout:
        if l >= 1:
            pos.extend(range(cur, cur + l))
            cur += l
        else:
            cur += 1
    return np.array(pos)


def mask_word(w, params):
    """``mask_word`` (:1207-1215): each chosen word becomes <mask> / stays / becomes a random word with params.pred_probs."""
    rand = np.random.randint(params.n_words, size=w.shape)
    probs = torch.multinomial(params.pred_probs, len(w), replacement=True)
    return np.full(w.shape, params.mask_index) * (probs == 0).numpy() + w * (probs == 1).numpy() + rand * (probs == 2).numpy()


def _columns(rows, length, n, pad):
    out = torch.LongTensor(length, n).fill_(pad)
    for i, r in enumerate(rows):
        out[:len(r), i].copy_(torch.LongTensor(r))
    return out


def restricted_mask_sent(x, lengths, params, min_len=100000):
    """MASS batch (:1269-1316): round(shortest sentence * params.word_mass) words of every sentence are masked in the
    encoder input x1 (80 / 10 / 10 rule); the decoder reads the word BEFORE each masked one (x2), at its original position
    (pos), and predicts the masked word (y).  -> (x1, len1, x2, len2, y, pred_mask, pos)."""
    min_len = max(min_len, 1)
    n = lengths.size(0)
    mask_len = round(lengths[np.argsort(lengths)[0].item()].item() * params.word_mass)
    gaps = [0] * (lengths.min().item() - mask_len - 1)
    spans = span_lengths(mask_len, min_len)
    inputs, targets, outputs, positions = [], [], [], []
    for i in range(n):
        words = np.array(x[:lengths[i], i].tolist())
        pos = span_positions(place_spans(spans, gaps))
        outputs.append(words[pos].copy())
        targets.append(words[pos - 1].copy())
        words[pos] = mask_word(words[pos], params)
        inputs.append(words)
        positions.append(pos - 1)
    pad = params.pad_index
    x1 = _columns(inputs, int(max(lengths)), n, pad)
    x2, y, pos = (_columns(v, mask_len, n, pad) for v in (targets, outputs, positions))
    pred_mask = y != pad
    return x1, lengths.clone(), x2, torch.LongTensor([mask_len] * n), y.masked_select(pred_mask), pred_mask, pos


def bart_token_mask_sent(x, lengths, params, min_len=100000):
    """BART-style text infilling batch (:1318-1381): ONE span of  Poisson(3) mod round(0.3 * slen)  words (at least one) per
    sentence collapses into a single <mask> in the encoder input x1; the decoder is teacher-forced on the whole original
    sentence (x2 = all but the last symbol, y = all but the first).  -> (x1, len1, x2, len2, y, pred_mask, pos)."""
    min_len = max(min_len, 1)
    n = lengths.size(0)
    mask_len = np.random.poisson(lam=3) % round(len(x[:, 0]) * 0.3)
    if mask_len == 0:
        mask_len = 1
    len1 = [lengths[i] - mask_len + 1 for i in range(n)]
    len2 = [lengths[i] - 1 for i in range(n)]
    gaps = [0] * (lengths.min().item() - mask_len - 1)
    spans = span_lengths(mask_len, min_len)
    inputs, targets, outputs, positions = [], [], [], []
    for i in range(n):
        words = np.array(x[:lengths[i], i].tolist())
        pos = span_positions(place_spans(spans, gaps))
        kept = np.concatenate([words[:pos[0]], words[pos[-1]:]])       # the span's last word stays, its first slot ...
        kept[pos[0]] = params.mask_index                               # ... becomes the <mask>
        inputs.append(kept)
        targets.append(words[:-1].copy())
        outputs.append(words[1:].copy())
        positions.append(np.arange(len(words) - 1))
    pad = params.pad_index
    x1 = _columns(inputs, int(max(len1)), n, pad)
    x2, y, pos = (_columns(v, int(max(len2)), n, pad) for v in (targets, outputs, positions))
    pred_mask = y != pad
    return x1, torch.LongTensor(len1), x2, torch.LongTensor(len2), y.masked_select(pred_mask), pred_mask, pos


# ---- region-feature noise of the image denoising step (xtrainer.py:1699-1744) -------------------------------------------------

def mask_object(object_features, mask_len=50):
    """``_mask_object`` (:1699-1732) on one image's region features (R, 2048) numpy: with probability 0.15 a region is hit;
    nine times in ten it becomes a zero vector that stands for itself AND the next  Poisson(3) mod mask_len  regions (a span
    collapses into one blank), otherwise it stays.  The result is cut / zero-padded to R - mask_len regions and every row is
    L2-normalised again (zero rows stay zero).  RNG order: one np.random.poisson, then one random.random() per visited region."""
    import random
    import torch.nn.functional as F
    rows = []
    max_len = len(object_features) - mask_len
    span = np.random.poisson(lam=3) % mask_len
    i = 0
    while i < len(object_features):
        prob = random.random()
        if prob < 0.15 and prob / 0.15 < 0.9:
            rows.append(np.zeros((2048), dtype=np.float32))
            i += span
        else:
            rows.append(object_features[i])
        i += 1
    rows = rows[:max_len] + [np.zeros((2048), dtype=np.float32)] * max(max_len - len(rows), 0)
    return F.normalize(torch.FloatTensor(np.stack(rows, 0)), dim=-1).numpy()


def bart_img_noise(object_features, loc_features, img_mask):
    """``bart_img_noise`` (:1734-1744): every image of the batch through ``mask_object`` with one common
    mask_len = Poisson(3) mod round(R / 2) + 1; boxes and mask are cut to the shortened region count."""
    feats = object_features.numpy()
    mask_len = np.random.poisson(lam=3) % (round(len(object_features[0]) * 0.5)) + 1
    out = torch.FloatTensor(np.stack([mask_object(feats[i], mask_len) for i in range(len(feats))], 0))
    n = out.shape[1]
    return out, loc_features[:, :n], img_mask[:, :n]
