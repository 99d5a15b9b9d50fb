"""Collate functions of the pre-training / retrieval data loaders: what turns per-item dataset
tuples into the batch tuples ``pretrain_under_step`` / ``t2i_step`` / ``i2t_step`` consume
(M3P/src/xtrainer.py:829-880 ``batch_sentences(_v2)``, :883-930 ``retrieval_collate``,
:960-1045 ``retrieval_pretrain_collate``), and those of the generation loaders that feed ``ic_step`` /
``mt_ic_step`` / the text-to-text steps (:931-957 ``caption_collate``, :1048-1075 ``mt_caption_collate``,
:1078-1087 ``ntg_collate``, :1089-1125 ``slide_collate``).  Host-side tensor packing only.

Conventions kept from the reference: sentences become (slen, n) int64 with BOS = 0 in row 0,
EOS = 2 after the last token and PAD = 1 below; language ids fill a (slen, n) tensor that
starts at 4; MLM labels are -1 wherever nothing is predicted (BOS / EOS / pad included).
Each dataset item holds ``sample_n`` captions for one image-tensor group: per-item tensors
are stacked and their two leading dimensions merged, per-item lists are concatenated."""
import numpy as np
import torch

BOS, PAD, EOS = 0, 1, 2


def _pack(sentences, fill):
    lengths = torch.LongTensor([len(s) + 2 for s in sentences])
    return lengths, torch.full((int(lengths.max()), len(sentences)), fill, dtype=torch.long)


def _put(col, n, values):
    """col[1 : n-1] = values for a sentence of n rows including BOS / EOS (nothing for an empty one)."""
    if n > 2:
        col[1:n - 1] = torch.from_numpy(np.asarray(values).astype(np.int64))


def batch_sentences(sentences, lg_ids=None):
    """xtrainer.py:829-852 -> (sent, lengths[, langs])."""
    lengths, sent = _pack(sentences, PAD)
    sent[0] = BOS
    for i, s in enumerate(sentences):
        n = int(lengths[i])
        _put(sent[:, i], n, s)
        sent[n - 1, i] = EOS
    if lg_ids is None:
        return sent, lengths
    langs = torch.full_like(sent, 4)
    for i in range(len(sentences)):
        langs[:, i] = lg_ids[i]
    return sent, lengths, langs


def batch_sentences_v2(sentences, lm_labels=None):
    """xtrainer.py:855-880 -> (sent, lengths[, labels]); labels -1 = not predicted."""
    lengths, sent = _pack(sentences, PAD)
    labels = torch.full_like(sent, -1) if lm_labels is not None else None
    sent[0] = BOS
    for i, s in enumerate(sentences):
        n = int(lengths[i])
        _put(sent[:, i], n, s)
        sent[n - 1, i] = EOS
        if labels is not None:
            _put(labels[:, i], n, lm_labels[i])
    if labels is None:
        return sent, lengths
    return sent, lengths, labels


def _merge(per_item):
    """Per-item tensors (k, ...) -> one (n_items * k, ...) tensor."""
    t = torch.stack(per_item, dim=0)
    return t.view([-1] + list(t.shape[2:]))


def _chain(per_item):
    out = []
    for x in per_item:
        out.extend(x)
    return out


def retrieval_collate(data):
    """Fine-tuning loader (xtrainer.py:883-930).  Each element of ``data`` is a pair (t2i item, i2t item); an item
    is (captions, region feats, region mask, box feats, object labels, positive index per group, image ids,
    language ids per caption).  Returns [t2i_batch, i2t_batch], each
    [(sent, lengths, langs), [img, img_mask, img_loc, obj_labels, pos_labels, img_ids]]."""
    def one(items):
        sent, feats, masks, boxes, objs, pos, ids, langs = zip(*items)
        return [batch_sentences(_chain(sent), _chain(langs)),
                [_merge(feats), _merge(masks), _merge(boxes), _merge(objs), _chain(pos), _chain(ids)]]
    t2i, i2t = zip(*data)
    return [one(t2i) if t2i is not None else None, one(i2t) if i2t is not None else None]


def retrieval_pretrain_collate(data):
    """Pre-training loader (xtrainer.py:960-1045).  t2i item = (captions, region feats, region mask, box feats,
    object labels, MLM labels per caption, ITM label, image ids, original region feats, masked types); the i2t
    item carries two more fields (CLCM captions, CLCM labels).  Returns [t2i_batch, i2t_batch] with
    t2i = [(sent, lengths, labels), [img, img_mask, img_loc, obj_labels, itm_labels, ori_feats, img_ids]] and
    i2t = [(sent, lengths, labels), (sent2, lengths2), [clcm_labels, img, ... as t2i]]."""
    def visual(feats, masks, boxes, objs, itm, ori, ids):
        return [_merge(feats), _merge(masks), _merge(boxes), _merge(objs), itm, _merge(ori), _chain(ids)]

    def t2i_side(items):
        sent, feats, masks, boxes, objs, lm, itm, ids, ori, _types = zip(*items)
        return [batch_sentences_v2(_chain(sent), _chain(lm)), visual(feats, masks, boxes, objs, itm, ori, ids)]

    def i2t_side(items):
        sent, feats, masks, boxes, objs, lm, itm, ids, ori, _types, sent2, clcm = zip(*items)
        return [batch_sentences_v2(_chain(sent), _chain(lm)), batch_sentences_v2(_chain(sent2), None),
                [torch.stack(clcm, dim=0)] + visual(feats, masks, boxes, objs, itm, ori, ids)]

    t2i, i2t = zip(*data)
    return [t2i_side(t2i) if t2i is not None else None, i2t_side(i2t) if i2t is not None else None]


def _regions(feats, masks, boxes):
    """Per-item (k, R, ...) region tensors -> (n_items * k, R, ...): features, mask, boxes."""
    return [_merge(feats), _merge(masks), _merge(boxes)]


def caption_collate(data):
    """Captioning loader (xtrainer.py:931-957).  Item = (captions, region feats, region mask, box feats, image ids); the
    captions of a batch are the per-item entries themselves (one word-id array per item, no chaining) and the image ids
    stay a tuple of per-item values.  Returns [(sent, lengths), [img, img_mask, img_loc, img_ids]]."""
    sent, feats, masks, boxes, ids = zip(*data)
    x_img, x_mask, loc = _regions(feats, masks, boxes)
    return [batch_sentences(sent), [x_img, x_mask, loc, ids]]


def mt_caption_collate(data):
    """Multimodal-translation loader (xtrainer.py:1048-1075).  Item = (source sentence, target sentence, region feats,
    region mask, box feats, image ids) -> [(src, src_len), (tgt, tgt_len), [img, img_mask, img_loc, img_ids]]."""
    src, tgt, feats, masks, boxes, ids = zip(*data)
    x_img, x_mask, loc = _regions(feats, masks, boxes)
    return [batch_sentences(src), batch_sentences(tgt), [x_img, x_mask, loc, ids]]


def ntg_collate(data):
    """Text-to-text generation loader (xtrainer.py:1078-1087): items (source, target) -> [(src, len), (tgt, len)]."""
    src, tgt = zip(*data)
    return [batch_sentences(src), batch_sentences(tgt)]


def slide_collate(data):
    """Sliding-window retrieval loader (xtrainer.py:1089-1125).  Item = (captions, region feats, region mask, box feats,
    image ids, labels) with per-item lists that are concatenated -> [(sent, lengths), [img, img_mask, img_loc, img_ids],
    labels]."""
    sent, feats, masks, boxes, ids, labels = zip(*data)
    x_img, x_mask, loc = _regions(feats, masks, boxes)
    return [batch_sentences(_chain(sent)), [x_img, x_mask, loc, _chain(ids)], _chain(labels)]
