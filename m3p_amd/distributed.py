"""Data parallelism for the hot path: one process per MI355X, RCCL over xGMI through
``torch.distributed`` (backend "nccl" *is* RCCL on ROCm).

Replaces the reference's wiring at M3P/src/xtrainer.py:66-83 (Apex
``DistributedDataParallel(delay_allreduce=True)`` — one flat all-reduce of every gradient
*after* backward, fully exposed) and M3P/src/slurm.py:156-170 (process-group init).

Design: the model's gradients already live in ONE flat fp32 arena laid out in forward
order, so a "bucket" is just an arena slice: the encoder backward (functional.EncoderFn)
calls ``layer_done(i)`` as soon as layer i's weight-gradient kernels are enqueued; the
reducer records an event on the compute stream, makes the communication stream wait on it
and launches the all-reduce(SUM) of that layer's ~28 MB slice there — overlapped with the
remaining layers' backward.  Embedding + head gradients (the 768 MB tied vocabulary matrix
is only final after the embedding scatter at the very end of backward) go last.
Averaging (1/world) is folded into the Adam kernel's ``grad_scale``; nothing is copied,
flattened or unflattened.  Unused reference parameters never enter a bucket, so ranks
always agree on the bucket plan.
"""
import os

import torch
import torch.distributed as dist


def init_distributed_mode(params=None, backend=None):
    """Process-group initialisation from the torchrun environment (slurm.py:156-170's
    ``init_process_group(init_method='env://', backend='nccl')`` without the SLURM parsing).
    Returns (rank, local_rank, world_size)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, init_method='env://')
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    if params is not None:
        params.global_rank, params.local_rank, params.world_size = rank, local_rank, world
        params.multi_gpu = world > 1
        params.n_gpu_per_node = world
        params.is_master = rank == 0
    return rank, local_rank, world


class BucketReducer:
    """Arena-slice all-reduce scheduler (works on any flat gradient tensor + list of
    [start, end) ranges, so it is testable on CPU with gloo)."""

    def __init__(self, flat_grad, process_group=None, use_side_stream=None):
        self.flat = flat_grad
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if use_side_stream is None:
            use_side_stream = flat_grad.is_cuda
        self.stream = torch.cuda.Stream() if use_side_stream else None
        self.pending = []
        self.enabled = True

    def reduce_range(self, start, end):
        if self.world == 1 or not self.enabled or end <= start:
            return
        buf = self.flat[start:end]
        if self.stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        else:
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self.pending.append(work)

    def finish(self):
        """Block the compute stream (not the host) until every launched bucket is reduced."""
        for w in self.pending:
            w.wait()
        self.pending = []
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)


class DataParallel(torch.nn.Module):
    """Wrapper with the surface the reference expects from (Apex) DDP: ``.module``,
    ``__call__(mode, **kw)`` forwarding, parameter broadcast from rank 0 at wrap time
    (xtrainer.py:68-83; used at :519-520, :814 and xevaluator.py:1532)."""

    def __init__(self, module, process_group=None, broadcast=True):
        super().__init__()
        self.module = module
        arena = module.arena()
        self.reducer = BucketReducer(arena.grad, process_group)
        self.world = self.reducer.world
        self._arena = arena
        self._embed_reduced = False
        module.ddp_hook = self
        if broadcast and self.world > 1:
            dist.broadcast(arena.master, src=0, group=process_group)
            for p in module.parameters():
                if getattr(p, '_m3p_arena', None) is None:
                    dist.broadcast(p.data, src=0, group=process_group)
            arena.mark_master_changed()

    def forward(self, mode, **kwargs):
        return self.module(mode, **kwargs)

    # ---- hooks called by functional.EncoderFn.backward
    def layer_done(self, i):
        s, e = self._arena.layer_ranges[i]
        self.reducer.reduce_range(s, e)

    def embed_done(self):
        s, e = self._arena.embed_range
        self.reducer.reduce_range(s, e)
        self._embed_reduced = True

    def no_sync(self):
        """Context manager: skip the collectives on non-boundary micro-steps of gradient
        accumulation (the reference all-reduces on every micro-step, xtrainer.py:231-243)."""
        red = self.reducer

        class _Ctx:
            def __enter__(self_):
                red.enabled = False

            def __exit__(self_, *a):
                red.enabled = True
        return _Ctx()

    def finish(self):
        """Called before clip/Adam: reduce what backward could not schedule (head gradients
        come from plain autograd accumulation) and wait for everything."""
        if self.world > 1 and self.reducer.enabled:
            if not self._embed_reduced:
                s, e = self._arena.embed_range
                self.reducer.reduce_range(s, e)
            s, e = self._arena.head_range
            self.reducer.reduce_range(s, e)
        self._embed_reduced = False
        self.reducer.finish()
