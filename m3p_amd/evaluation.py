"""Retrieval scoring on the MI355X model — the arithmetic of
``evaluate_image_retrieval`` (M3P/src/evaluation/xevaluator.py:1528-1657): every image is
scored against every caption with the cross-encoder (``jointfwd`` + ``predict(is_relation)``
under ``no_grad``), Recall@{1,5,10} is read off the score matrix by top-k (:1621-1657).
Differences from the reference, on purpose: the image is broadcast over a chunk of captions
by indexing instead of ``repeat`` (:1563-1564), scores stay on the device, images can be
sharded over ranks (the reference slices by ``local_rank``, dataset_finetune.py:1218-1219)."""
import torch


@torch.no_grad()
def relation_score_matrix(model, x, lengths, x_img, image_loc, chunk=64, rank=0, world=1):
    """x (T, n_cap) int64, lengths (n_cap,), x_img (R, n_img, 2048), image_loc (R, n_img, 5).
    Returns (scores [n_img_local, n_cap] fp32, image indices of this rank)."""
    was_training = model.training
    model.eval()
    dev = x.device
    n_cap, n_img = x.shape[1], x_img.shape[1]
    R = x_img.shape[0]
    mine = torch.arange(rank, n_img, world, device=dev)
    out = torch.empty((mine.numel(), n_cap), dtype=torch.float32, device=dev)
    img_len = torch.full((1,), R, dtype=torch.long, device=dev)
    for row, i in enumerate(mine.tolist()):
        for c0 in range(0, n_cap, chunk):
            c1 = min(n_cap, c0 + chunk)
            nb = c1 - c0
            xi = x_img[:, i:i + 1].expand(R, nb, x_img.shape[2]).contiguous()
            li = image_loc[:, i:i + 1].expand(R, nb, 5).contiguous()
            enc = model('jointfwd', x=x[:, c0:c1].contiguous(), lengths=lengths[c0:c1], x_img=xi,
                        lengths_img=img_len.expand(nb), causal=False, langs=None, image_loc=li, refine_image=False)
            out[row, c0:c1] = model('predict', tensor=enc.transpose(0, 1), is_relation=True).view(-1).float()
    if was_training:
        model.train()
    return out, mine


def recall_at_k(scores, gt, ks=(1, 5, 10)):
    """scores [n_query, n_cand]; gt[q] = index of the ground-truth candidate (or a bool mask
    [n_query, n_cand] when several candidates are correct, e.g. 5 captions per image)."""
    order = torch.argsort(scores, dim=1, descending=True)
    if gt.dim() == 1:
        rank = (order == gt[:, None]).float().argmax(dim=1)
    else:
        hit = torch.gather(gt, 1, order)
        rank = hit.float().argmax(dim=1)
    return {k: float((rank < k).float().mean()) for k in ks}
