"""Data-parallel wrapper on the real model: two ranks (gloo, both on cuda:0 — one MI355X is
all a test box has; RCCL itself is exercised by the driver's multi-GPU bench) each take half
of a batch; the bucketed all-reduce scheduled from inside backward + 1/world in the optimizer
must reproduce the single-process gradient of the whole batch, and parameters stay identical."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from m3p_amd import synth

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _half(batch, r, world):
    B = batch['x'].shape[1]
    per = B // world
    sl = slice(r * per, (r + 1) * per)
    out = {k: (v[:, sl].contiguous() if k in ('x', 'x_labels', 'pred_mask', 'x_img', 'image_loc') else v)
           for k, v in batch.items()}
    out['lengths'] = batch['lengths'][sl].contiguous()
    out['lengths_img'] = batch['lengths_img'][sl].contiguous()
    out['y'] = out['x_labels'][out['pred_mask']]
    ng = per // 2
    out['pos_labels'] = batch['pos_labels'][r * ng:(r + 1) * ng]
    return out


def _loss(m, batch, R):
    dev = 'cuda'
    out = m('jointfwd', x=batch['x'].to(dev), lengths=batch['lengths'].to(dev), x_img=batch['x_img'].to(dev),
            lengths_img=batch['lengths_img'].to(dev), causal=False, langs=None, image_loc=batch['image_loc'].to(dev),
            refine_image=False)
    _, mlm = m('predict', tensor=out[R:], pred_mask=batch['pred_mask'].to(dev), y=batch['y'].to(dev), get_scores=False)
    rel = m('predict', tensor=out.transpose(0, 1), is_relation=True)
    onehot = torch.eye(2, device=dev)[batch['pos_labels'].to(dev)].reshape(-1)
    return mlm + torch.nn.functional.binary_cross_entropy_with_logits(rel.view(-1).float(), onehot)


def _build(cfg):
    from m3p_amd.model.transformer import TransformerModel
    P = synth.model_params(cfg['emb_dim'], cfg['n_heads'], cfg['n_layers'], cfg['n_words'])
    m = TransformerModel(P, is_encoder=True, with_output=True, is_crossModal=True)
    m.load_state_dict(synth.golden_state_dict(synth.hot_param_shapes(P)), strict=False)
    return m.cuda().train()


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.cuda.set_device(0)
        dist.init_process_group('gloo', rank=rank, world_size=world)
        from m3p_amd.distributed import DataParallel
        from m3p_amd.optim import get_optimizer
        cfg = dict(emb_dim=128, n_heads=4, n_layers=2, n_words=1000, T=24, R=10, B=8, n_pred=4)
        full = synth.make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'], seed=11, ragged=False)
        m = _build(cfg)
        ddp = DataParallel(m)
        opt = get_optimizer([p for p in m.parameters()], 'adam_inverse_sqrt,beta1=0.9,beta2=0.98,lr=0.0001')
        opt.grad_scale = 1.0 / world
        loss = _loss(ddp, _half(full, rank, world), cfg['R'])
        loss.backward()
        ddp.finish()
        torch.cuda.synchronize()
        g = (m.arena().grad / world).cpu()
        opt.clip_grad_norm(5.0)
        opt.step()
        torch.cuda.synchronize()
        pm = m.arena().master.cpu()
        gathered = [torch.zeros_like(pm) for _ in range(world)]
        dist.all_gather(gathered, pm)
        same = all(torch.equal(gathered[0], t) for t in gathered)
        if rank == 0:
            ref = _build(cfg)
            _loss(ref, full, cfg['R']).backward()
            torch.cuda.synchronize()
            gr = ref.arena().grad.cpu()
            err = float((g - gr).norm() / gr.norm())
            q.put(('ok', err, same, float(m.arena().grad.abs().max())))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:   # surface the reason to the parent
        if rank == 0:
            q.put(('err', repr(e), False, 0.0))
        raise


def test_dp_two_ranks_match_single_process():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    status, err, same, gmax = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
    assert status == 'ok', err
    assert err < 2e-2, err
    assert same, 'parameters diverged across ranks'
    assert gmax == 0.0      # fused zero_grad
