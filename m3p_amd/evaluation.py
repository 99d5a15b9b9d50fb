"""Retrieval scoring on the MI355X model - the arithmetic of ``evaluate_image_retrieval``
(M3P/src/evaluation/xevaluator.py:1528-1657): every image is scored against every caption with
the cross-encoder (``jointfwd`` + ``predict(is_relation)`` under ``no_grad``), then
Recall@{1,5,10} is read off the (n_img, n_cap) score matrix in both directions (:1621-1657).

Differences from the reference, on purpose:
  * the image broadcast is an index gather on the device: one encoder call covers a tile of
    ``img_block`` images x ``chunk`` captions (the reference ``repeat``s one retrieval batch of
    images over a caption split, :1563-1564, and loops in Python);
  * scores stay on the device until the metric is read;
  * images are sharded over ranks by stride (the reference slices the image list by
    ``local_rank``, dataset_finetune.py:1218-1219) and the shards can be all-gathered.

Valid-set matching accuracy (``evaluate_t2i`` / ``evaluate_i2t`` / ``evaluate_understanding_tasks``,
xevaluator.py:1262-1417): each dataset item contributes ``sample_n`` (text, image) sequences, one of them the true
pair; a group counts as correct when its highest relation score sits on ``pos_labels``.
"""
import numpy as np
import torch
import torch.distributed as dist

from .utils import to_cuda


@torch.no_grad()
def relation_score_matrix(model, x, lengths, x_img, image_loc, chunk=256, img_block=4, rank=0, world=1, refine_image=False):
    """x (T, n_cap) int64, lengths (n_cap,), x_img (R, n_img, 2048), image_loc (R, n_img, 5), all on the device.
    Returns (scores [n_img_local, n_cap] fp32, image indices of this rank)."""
    was_training = model.training
    model.eval()
    dev = x.device
    n_cap, n_img = x.shape[1], x_img.shape[1]
    R = x_img.shape[0]
    mine = torch.arange(rank, n_img, world, device=dev)
    out = torch.empty((mine.numel(), n_cap), dtype=torch.float32, device=dev)
    for r0 in range(0, mine.numel(), img_block):
        imgs = mine[r0:r0 + img_block]
        ni = imgs.numel()
        for c0 in range(0, n_cap, chunk):
            nc = min(n_cap, c0 + chunk) - c0
            # sequence j of the tile = (image j // nc, caption c0 + j % nc)
            img_of = imgs.repeat_interleave(nc)
            xi = x_img.index_select(1, img_of)
            li = image_loc.index_select(1, img_of)
            xt = x[:, c0:c0 + nc].repeat(1, ni)
            lt = lengths[c0:c0 + nc].repeat(ni)
            enc = model('jointfwd', x=xt, lengths=lt, x_img=xi, lengths_img=torch.full((ni * nc,), R, dtype=torch.long, device=dev),
                        causal=False, langs=None, image_loc=li, refine_image=refine_image)
            s = model('predict', tensor=enc.transpose(0, 1), is_relation=True)
            out[r0:r0 + ni, c0:c0 + nc] = s.view(ni, nc).float()
    if was_training:
        model.train()
    return out, mine


def gather_score_matrix(local_scores, mine, n_img, group=None):
    """All ranks' shards -> the full (n_img, n_cap) matrix on every rank (shards are strided: image i lives on rank
    i % world; short shards are padded to the longest for the collective)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_scores
    n_cap = local_scores.shape[1]
    per = (n_img + world - 1) // world
    buf = torch.zeros((per, n_cap), dtype=local_scores.dtype, device=local_scores.device)
    buf[:local_scores.shape[0]] = local_scores
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    full = torch.empty((n_img, n_cap), dtype=local_scores.dtype, device=local_scores.device)
    for r in range(world):
        idx = torch.arange(r, n_img, world, device=local_scores.device)
        full[idx] = parts[r][:idx.numel()]
    return full


def recall_at_k(scores, gt, ks=(1, 5, 10)):
    """scores [n_query, n_cand]; gt[q] = index of the ground-truth candidate (or a bool mask
    [n_query, n_cand] when several candidates are correct, e.g. 5 captions per image):
    fraction of queries with a correct candidate among the K best."""
    order = torch.argsort(scores, dim=1, descending=True)
    if gt.dim() == 1:
        rank = (order == gt[:, None]).float().argmax(dim=1)
    else:
        hit = torch.gather(gt, 1, order)
        rank = hit.float().argmax(dim=1)
    return {k: float((rank < k).float().mean()) for k in ks}


def retrieval_recalls(scores, labels):
    """xevaluator.py:1621-1657 on an (n_img, n_cap) score matrix and its 0/1 label matrix, vectorised:
    -> (t2i_r1, t2i_r5, t2i_r10, i2t_r1, i2t_r5, i2t_r10).  image -> sentence counts, per image, the first positive
    among its 10 best captions; sentence -> image counts, per caption, every positive among its 10 best images (the
    reference loop has no early exit there) and divides by the number of captions."""
    n_img, n_cap = scores.shape
    lab = labels.to(scores.device) == 1
    out = []
    # sentence -> image
    top = scores.t().topk(min(10, n_img), dim=-1).indices            # (n_cap, 10) image ids
    hit = torch.gather(lab.t(), 1, top)
    out += [float(hit[:, :k].sum()) / n_cap for k in (1, 5, 10)]
    # image -> sentence
    top = scores.topk(min(10, n_cap), dim=-1).indices                # (n_img, 10) caption ids
    hit = torch.gather(lab, 1, top)
    first = torch.where(hit.any(dim=1), hit.float().argmax(dim=1), torch.full((n_img,), 10**6, device=scores.device))
    out += [float((first < k).sum()) / n_img for k in (1, 5, 10)]
    return tuple(out)


@torch.no_grad()
def _matching_accuracy(model, params, batch):
    """One collate batch -> (#groups whose best-scoring member is the labelled pair, #groups).  Both layouts the
    reference unpacks: the pre-training tuples (t2i: (x1, len1, labels), (img, mask, loc, obj, pos, ori, ids); i2t with
    the extra (x2, len2) and CLCM labels) and the fine-tuning one ((x1, len1, lang_p), (img, mask, loc, pos, ids)).
    The per-position language ids the reference assembles (:1317-1326) feed ``langs``, which jointfwd ignores
    (transformer.py:937-938) - not built."""
    model = getattr(model, 'module', model)
    was_training = model.training
    model.eval()
    text, visual = batch[0], batch[-1]
    x1, len1 = text[0], text[1]
    if getattr(params, 'is_pretrain', False):
        if len(batch) == 3:                 # i2t: (clcm_labels, img, img_mask, img_loc, obj_labels, pos_labels, img_ori, img_ids)
            img, img_mask, img_loc, pos_labels = visual[1], visual[2], visual[3], visual[5]
        else:                               # t2i: (img, img_mask, img_loc, obj_labels, pos_labels, img_ori, img_ids)
            img, img_mask, img_loc, pos_labels = visual[0], visual[1], visual[2], visual[4]
    else:                                   # (img, img_mask, img_loc, pos_labels, img_ids); retrieval_collate adds obj_labels
        img, img_mask, img_loc = visual[0], visual[1], visual[2]
        pos_labels = visual[4] if len(visual) == 6 else visual[3]
    img_len = img_mask.sum(dim=1)
    x1, len1, x_img, loc, img_len = to_cuda(x1, len1, img.transpose(0, 1), img_loc.transpose(0, 1), img_len)
    enc = model('jointfwd', x=x1, lengths=len1, x_img=x_img, lengths_img=img_len, causal=False, langs=None,
                image_loc=loc, refine_image=getattr(params, 'refine_image', False))
    scores = model('predict', tensor=enc.transpose(0, 1), is_relation=True)
    pred = scores.view(-1, params.sample_n).float().argmax(dim=1).cpu()         # the step's one device read
    label = torch.from_numpy(np.asarray(pos_labels)).reshape(-1)
    if was_training:
        model.train()
    return int((pred == label).sum()), int(label.numel())


def evaluate_t2i(model, params, batch):
    """xevaluator.py:1309-1359."""
    return _matching_accuracy(model, params, batch)


def evaluate_i2t(model, params, batch):
    """xevaluator.py:1361-1417 (the same arithmetic on the i2t tuple)."""
    return _matching_accuracy(model, params, batch)


def evaluate_understanding_tasks(model, params, iterator, scores, data_set, lang1, lang2):
    """xevaluator.py:1262-1307: accumulates both accuracies over ``iterator`` (pairs of (t2i_batch, i2t_batch), what
    ``get_iterator(data_set, lang1, lang2)`` yields there) into ``scores`` under the reference's keys."""
    assert data_set in ('valid', 'test')
    t2i_acc = t2i_n = i2t_acc = i2t_n = 0
    for t2i_batch, i2t_batch in iterator:
        a, n = evaluate_t2i(model, params, t2i_batch)
        t2i_acc, t2i_n = t2i_acc + a, t2i_n + n
        a, n = evaluate_i2t(model, params, i2t_batch)
        i2t_acc, i2t_n = i2t_acc + a, i2t_n + n
    scores['%s_%s-%s_rel_t2i_acc' % (data_set, lang1, lang2)] = 100. * t2i_acc / t2i_n
    scores['%s_%s-%s_rel_i2t_acc' % (data_set, lang1, lang2)] = 100. * i2t_acc / i2t_n
    return scores
