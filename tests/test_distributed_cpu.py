"""Data-parallel path on CPU (gloo, world_size 2): the arena-slice reducer used by
m3p_amd.distributed.DataParallel sums bucket ranges in place, and the DP recipe
(per-rank mean loss -> all-reduce(SUM) -> 1/world inside the optimizer) reproduces the
single-process gradient of the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from m3p_amd import synth


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat_grads(sd, names, cfg, batch):
    from oracle import ref_cpu as O
    leaves = {n: sd[n].clone().requires_grad_(True) for n in names}
    res = O.pretrain_losses(leaves, cfg['n_layers'], cfg['n_heads'], batch, cfg['R'])
    grads = torch.autograd.grad(res['total'], [leaves[n] for n in names])
    return torch.cat([g.reshape(-1) for g in grads]), float(res['total'])


def _half(batch, r, world):
    B = batch['x'].shape[1]
    per = B // world
    sl = slice(r * per, (r + 1) * per)
    out = {}
    for k, v in batch.items():
        if k in ('x', 'x_labels', 'pred_mask'):
            out[k] = v[:, sl].contiguous()
        elif k in ('x_img', 'image_loc'):
            out[k] = v[:, sl].contiguous()
        elif k in ('lengths', 'lengths_img'):
            out[k] = v[sl].contiguous()
    out['y'] = out['x_labels'][out['pred_mask']]
    n_groups = per // 2
    out['pos_labels'] = batch['pos_labels'][r * n_groups:(r + 1) * n_groups]
    return out


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from m3p_amd.distributed import BucketReducer
    cfg = dict(emb_dim=64, n_heads=2, n_layers=2, n_words=200, T=12, R=4, B=8, n_pred=2)
    P = synth.model_params(cfg['emb_dim'], cfg['n_heads'], cfg['n_layers'], cfg['n_words'])
    sd = synth.golden_state_dict(synth.hot_param_shapes(P))
    names = list(sd.keys())
    full = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'], seed=3, ragged=False)
    flat, loss = _flat_grads(sd, names, cfg, _half(full, rank, world))
    flat = flat.contiguous()
    red = BucketReducer(flat, use_side_stream=False)
    n = flat.numel()
    cuts = [0, n // 5, n // 2, n]                      # three "buckets", launched in reverse like backward does
    for a, b in reversed(list(zip(cuts[:-1], cuts[1:]))):
        red.reduce_range(a, b)
    red.finish()
    avg = flat / world
    if rank == 0:
        ref, ref_loss = _flat_grads(sd, names, cfg, full)
        err = float((avg - ref).norm() / ref.norm())
        q.put((err, loss, ref_loss))
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_reducer_dp_equals_single_process():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    err, loss, ref_loss = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err < 1e-5, err          # fp32 reduction-order noise only


def test_reducer_single_process_is_noop():
    from m3p_amd.distributed import BucketReducer
    t = torch.arange(10, dtype=torch.float32)
    red = BucketReducer(t, use_side_stream=False)
    red.reduce_range(0, 10)
    red.finish()
    assert torch.equal(t, torch.arange(10, dtype=torch.float32))
