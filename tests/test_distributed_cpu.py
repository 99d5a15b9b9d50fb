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


# ---- the DataParallel step protocol on a fake arena (gloo, CPU): what functional.py / optim.py call, in order ----
class _FakeArena:
    """Just the bookkeeping m3p_amd.distributed.DataParallel reads from functional.Arena: buckets on 512-element
    boundaries (vocabulary matrix + bias | positions | layers | heads)."""

    def __init__(self, V=50, d=8, n_layers=2):
        from collections import OrderedDict
        self.offsets = OrderedDict([('embeddings.weight', (0, V * d, (V, d))), ('pred_layer.proj.bias', (448, 64, (64,))),
                                    ('position_embeddings.weight', (512, 128, (128,)))])
        off = 1024
        self.layer_ranges = []
        for i in range(n_layers):
            self.offsets['layer%d' % i] = (off, 192, (192,))
            self.layer_ranges.append((off, off + 512))
            off += 512
        self.offsets['pooled_layer.dense.weight'] = (off, 64, (64,))
        self.head_range = (off, off + 512)
        self.total = off + 512
        self.embed_range = (0, 1024)
        self.device = torch.device('cpu')
        self.master = torch.zeros(self.total)
        self.grad = torch.zeros(self.total)
        self.w16 = torch.zeros(self.total, dtype=torch.bfloat16)
        self.V, self.d = V, d
        self._transposes_stale = False
        self.stale = None                    # (start, count) of a lazily zeroed range, functional.Arena.defer_vocab_zero

    def ensure_zero(self):
        if self.stale is not None:
            self.grad[self.stale[0]:self.stale[0] + self.stale[1]].zero_()
            self.stale = None

    def g(self, name):
        o, n, _ = self.offsets[name]
        if self.stale is not None and o < self.stale[0] + self.stale[1] and o + n > self.stale[0]:
            self.ensure_zero()
        return self.grad[o:o + n].view(self.V, self.d) if name == 'embeddings.weight' else self.grad[o:o + n]

    def touch(self, *names):
        pass

    def mark_master_changed(self):
        pass

    def refresh_transposes(self):
        pass


class _FakeModel(torch.nn.Module):
    pad_index = 1

    def __init__(self):
        super().__init__()
        self._arena = _FakeArena()
        self.ddp_hook = None

    def arena(self):
        return self._arena


def _cpu_scatter(rows, ids, dst, pad_index):
    keep = ids != pad_index
    dst.index_add_(0, ids[keep], rows[keep].float())


def _protocol_worker(rank, world, port, q, mode):
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.set_num_threads(1)
        dist.init_process_group('gloo', rank=rank, world_size=world)
        from m3p_amd import ops
        from m3p_amd.distributed import DataParallel
        ops.scatter_add_token_rows = _cpu_scatter
        model = _FakeModel()
        ar = model.arena()
        ar.master.fill_(float(rank + 1))
        dp = DataParallel(model, mode=mode)
        assert dp.mode == mode
        assert float(ar.master[0]) == 1.0                       # broadcast from rank 0
        results = {}

        def fill(val):
            ar.grad.fill_(val)

        def tokens(n, seed):
            g = torch.Generator().manual_seed(seed)
            ids = torch.randint(0, ar.V, (n,), generator=g)
            ids[0] = model.pad_index                            # a pad token: its row must be ignored
            rows = torch.randn(n, ar.d, generator=g).to(torch.bfloat16)
            return ids, rows

        def expected_tokens(specs):
            out = torch.zeros(ar.V, ar.d)
            for n, seed in specs:
                ids, rows = tokens(n, seed)
                _cpu_scatter(rows, ids, out, model.pad_index)
            return out

        def reduced(launched_only_check=None):
            """what the optimizer would see: the reduced gradient, every rank's shard in its place"""
            own = torch.zeros(ar.total, dtype=torch.bool)
            for a, b in dp.owned(0, ar.total):
                own[a:b] = True
            full = dp.full_reduced_grad()
            return full, own

        # the shards of a bucket tile it exactly, across ranks
        cover = torch.zeros(ar.total)
        for a, b in dp.owned(0, ar.total):
            cover[a:b] += 1
        dist.all_reduce(cover)
        assert bool((cover == (1 if mode == 'zero1' else world)).all())

        # --- step A: one encoder pass + MLM head, finish() called twice (clip, then step); ragged token counts
        dp.plan_step(True)
        n_loc = 5 + 2 * rank
        n_max = dp.encoder_forward(n_loc)                       # still in flight: read in embed_done
        fill(1.0 + rank)
        dp.mlm_head_done()
        last = dp.encoder_backward_begin()
        assert last
        for i in (1, 0):
            dp.layer_done(i, last)
        ids, rows = tokens(n_loc, 100 + rank)
        dp.embed_done(last, ids=ids, rows=rows, n_max=n_max)
        dp.finish()
        dp.finish()                                             # idempotent: nothing is reduced twice
        tot = sum(1.0 + r for r in range(world))
        want = torch.full_like(ar.grad, tot)
        tokw = torch.zeros_like(ar.grad)
        tokw[:ar.V * ar.d] = expected_tokens([(5 + 2 * r, 100 + r) for r in range(world)]).view(-1)
        full, own = reduced()
        # (token rows are scatter-added on every rank after the reduction: a gathered shard carries them once)
        results['A'] = float((full - want - tokw).abs().max())
        if mode == 'zero1':
            # outside its shards a rank still holds its own partial sums, never the reduced value - and of the gathered token
            # rows only those whose matrix row reaches into its shard (a row cut by the shard boundary is added whole)
            extra = ar.grad[~own] - (1.0 + rank)
            assert bool(((extra.abs() < 1e-5) | ((extra - tokw[~own]).abs() < 1e-5)).all())
            if world > 1:
                assert float(extra.abs().sum()) < float(tokw[~own].abs().sum())      # (most foreign rows were skipped)
        dp.step_done()

        # --- step B: two encoder passes (CLCM): the first backward must not launch layer buckets
        dp.plan_step(True)
        nm1, nm2 = dp.encoder_forward(4), dp.encoder_forward(6)
        fill(0.0)
        assert not dp.encoder_backward_begin()                  # pass 2's backward comes first and is not the last
        ar.grad[ar.layer_ranges[1][0]:ar.layer_ranges[1][1]] += 1.0 + rank
        dp.layer_done(1, False)
        assert not dp._launched, 'a non-final backward launched a bucket'
        i2, r2 = tokens(6, 300 + rank)
        dp.embed_done(False, ids=i2, rows=r2, n_max=nm2)
        dp.mlm_head_done()
        assert dp.encoder_backward_begin()
        ar.grad[ar.layer_ranges[1][0]:ar.layer_ranges[1][1]] += 10.0
        dp.layer_done(1, True)
        dp.layer_done(0, True)
        i1, r1 = tokens(4, 200 + rank)
        dp.embed_done(True, ids=i1, rows=r1, n_max=nm1)
        dp.finish()
        want = torch.zeros_like(ar.grad)
        want[ar.layer_ranges[1][0]:ar.layer_ranges[1][1]] = tot + 10.0 * world
        want[:ar.V * ar.d] += expected_tokens([(6, 300 + r) for r in range(world)] + [(4, 200 + r) for r in range(world)]).view(-1)
        results['B'] = float((reduced()[0] - want).abs().max())
        dp.step_done()

        # --- step C: gradient accumulation: a no_sync micro-step keeps its token rows for the boundary; no MLM head
        dp.plan_step(False)
        fill(0.0)
        with dp.no_sync():
            nm = dp.encoder_forward(3)
            assert dp.encoder_backward_begin()
            ar.grad[ar.layer_ranges[0][0]:ar.layer_ranges[0][1]] += 1.0
            dp.layer_done(0, True)
            ia, ra = tokens(3, 400 + rank)
            dp.embed_done(True, ids=ia, rows=ra, n_max=nm)
            dp.finish()                                         # disabled: must not mark the step finished
        assert not dp._launched and len(dp._tokens) == 1
        nm = dp.encoder_forward(3)
        assert dp.encoder_backward_begin()
        ar.grad[ar.layer_ranges[0][0]:ar.layer_ranges[0][1]] += 1.0
        dp.layer_done(0, True)
        ib, rb = tokens(3, 500 + rank)
        dp.embed_done(True, ids=ib, rows=rb, n_max=nm)
        dp.finish()
        assert 'vocab' not in dp._launched                      # ITM-only step: the 768-MB bucket is never reduced
        want = torch.zeros_like(ar.grad)
        want[ar.layer_ranges[0][0]:ar.layer_ranges[0][1]] = 2.0 * world
        want[:ar.V * ar.d] += expected_tokens([(3, 400 + r) for r in range(world)] + [(3, 500 + r) for r in range(world)]).view(-1)
        results['C'] = float((reduced()[0] - want).abs().max())
        dp.step_done()

        # --- step D: no encoder backward at all (finish() owes every bucket)
        dp.plan_step(True)
        fill(2.0)
        dp.finish()
        results['D'] = float((reduced()[0] - 2.0 * world).abs().max())
        dp.step_done()

        # --- step E: an image-stream pass feeds the encoder pass: 'embed' waits for the stream's backward
        dp.plan_step(False)
        fill(0.0)
        dp.stream_forward()
        nm = dp.encoder_forward(0)
        assert dp.encoder_backward_begin()
        dp.layer_done(1, True); dp.layer_done(0, True)
        dp.embed_done(True, ids=None, rows=None, n_max=nm)
        assert 'embed' not in dp._launched
        ar.grad[512:640] += 3.0 + rank                          # what the image stream's backward writes (inside 'embed')
        dp.stream_backward_end()
        assert 'embed' in dp._launched
        dp.finish()
        want = torch.zeros_like(ar.grad)
        want[512:640] = sum(3.0 + r for r in range(world))
        results['E'] = float((reduced()[0] - want).abs().max())
        dp.step_done()

        # --- step F: the sharded optimizer step: every rank updates its shards only; afterwards all hold the same master
        dp.plan_step(True)
        fill(1.0)
        dp.finish()
        ar.master.fill_(-1.0)
        want = torch.arange(ar.total, dtype=torch.float32)
        dp._MATRIX_MIN_ELEMS = 256            # (this arena's 50 x 8 "vocabulary matrix" is to count as a sharded GEMM operand)
        for a, b in dp.owned(0, ar.total):
            ar.master[a:b] = want[a:b]                                    # "Adam" on the shard: master and bf16 copy together
            ar.w16[a:b] = want[a:b].to(torch.bfloat16)
        took = dp.after_sharded_step([(0, ar.total)])
        assert took == (mode == 'zero1')
        if mode == 'zero1':
            # the bf16 working copy is whole on every rank; of the fp32 master, everything but the big matrix
            # (embeddings.weight in this arena) is current everywhere, the matrix only where this rank owns it ...
            results['F16'] = float((ar.w16.float() - want.to(torch.bfloat16).float()).abs().max())
            vec = torch.ones(ar.total, dtype=torch.bool)
            o, n, _ = ar.offsets['embeddings.weight']
            vec[o:o + n] = False
            named = torch.zeros(ar.total, dtype=torch.bool)
            for _, (po, pn, _) in ar.offsets.items():
                named[po:po + pn] = True
            results['Fvec'] = float((ar.master - want)[vec & named].abs().max())
            own = torch.zeros(ar.total, dtype=torch.bool)
            for a, b in dp.owned(0, ar.total):
                own[a:b] = True
            stale = bool(((ar.master == -1.0) & ~own & ~vec & named).any())      # (what nobody sent: the matrix's foreign shards)
            results['Fstale'] = 0.0 if (stale or world == 1) else 1.0
            assert dp.master_partial
            # ... until every rank asks for the rest
            dp.materialize_master()
            assert not dp.master_partial
            results['F'] = float((ar.master - want)[named | own].abs().max())
            results['Fz'] = float(ar.grad[~reduced()[1]].abs().max())     # un-reduced partials outside the shards: zeroed
        dp.step_done()

        # --- step G (round 6): the optimizer left the vocabulary range lazily un-zeroed (it still holds last step's gradient, 7.0
        # here).  Rank 0's MLM head STORES over it; the other ranks' batches held no masked word - no MLM head, no store - and
        # their stale range must be zeroed before the 'vocab' bucket sums it with rank 0's.  Token rows enter through g().
        dp.plan_step(True)
        fill(0.0)
        v0, vc = 0, 512
        ar.grad[v0:v0 + vc] = 7.0
        ar.stale = (v0, vc)
        nm = dp.encoder_forward(2)
        if rank == 0:
            ar.grad[v0:v0 + vc] = 5.0                           # the store
            ar.stale = None
            dp.mlm_head_done()
        assert dp.encoder_backward_begin()
        dp.layer_done(1, True); dp.layer_done(0, True)
        ig, rg = tokens(2, 600 + rank)
        dp.embed_done(True, ids=ig, rows=rg, n_max=nm)
        dp.finish()
        assert ar.stale is None
        want = torch.zeros_like(ar.grad)
        want[v0:v0 + vc] = 5.0
        want[:ar.V * ar.d] += expected_tokens([(2, 600 + r) for r in range(world)]).view(-1)
        results['G'] = float((reduced()[0] - want).abs().max())
        if mode == 'zero1':
            # ... and the sharded step leaves a lazily kept range alone outside this rank's shards, zeroes the rest
            before = ar.grad.clone()
            dp.after_sharded_step([(0, ar.total)], keep=(v0, v0 + vc))
            own = reduced()[1]
            keep = torch.zeros(ar.total, dtype=torch.bool)
            keep[v0:v0 + vc] = True
            results['Gkeep'] = float((ar.grad - before)[keep & ~own].abs().max()) if bool((keep & ~own).any()) else 0.0
            results['Gzero'] = float(ar.grad[~keep & ~own].abs().max())
        dp.step_done()

        if rank == 0:
            q.put(('ok', results))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        if rank == 0:
            q.put(('err', traceback.format_exc()))
        raise


@pytest.mark.parametrize('world,mode', [(2, 'zero1'), (4, 'zero1'), (2, 'allreduce')])
def test_data_parallel_step_protocol(world, mode):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_protocol_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    status, res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
    assert status == 'ok', res
    for k, err in res.items():
        assert err < 1e-5, (k, err)
