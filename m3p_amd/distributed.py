"""Data parallelism for the hot path: one process per MI355X, RCCL over xGMI through
``torch.distributed`` (backend "nccl" *is* RCCL on ROCm).

Replaces the reference's wiring at M3P/src/xtrainer.py:66-83 (Apex
``DistributedDataParallel(delay_allreduce=True)`` — one flat all-reduce of every gradient
*after* backward, fully exposed) and M3P/src/slurm.py:156-170 (process-group init).

Design.  The model's gradients live in ONE flat fp32 arena laid out in forward order, so a
"bucket" is an arena slice and nothing is copied, flattened or unflattened.  Per optimizer
step the collectives are issued on a side stream in this fixed, rank-independent order:

  1. ``vocab``   the tied vocabulary matrix + its bias (768 MB at V = 250 002): only its DENSE
                 part, the MLM-head weight gradient, which is final as soon as
                 ``MLMHeadFn.backward`` returns - the very first thing backward does.  The
                 largest collective therefore overlaps with the whole encoder backward.
                 Steps without an MLM head (ITM fine-tuning) skip it entirely.
  2. ``heads``   pooler / relation / region heads (a few MB), when the last encoder backward starts.
  3. ``layer i`` as soon as layer i's weight-gradient kernels are enqueued (28 MB at 768d).
  4. ``embed``   positions, embedding LayerNorm, image projection (+ refiner) after the
                 assembly backward.
  5. ``tokens``  the embedding-LOOKUP gradient, the only part of the vocabulary matrix that
                 is produced at the very end of backward, travels as what it is: <= B*T rows.
                 ``m3p_embed_assemble_bwd`` writes them as bf16 rows instead of scattering,
                 the ranks all-gather (ids, rows) and every rank scatter-adds all of them in
                 fp32 (``m3p_scatter_add_token_rows``).  50 MB per rank at B = 256 instead of
                 a second pass over 768 MB.  (bf16 on the wire, fp32 accumulation: the rows
                 are gradients of bf16 activations; stated in DESIGN.md §5.)

A step may run more than one encoder pass (the CLCM objective runs ``jointfwd`` twice,
xtrainer.py:2379-2393): passes are counted in forward, and only the LAST backward of a step
launches layer / embed buckets - earlier ones just accumulate.  ``finish()`` launches
whatever the plan still owes, waits, applies the token rows; it is idempotent per step (the
optimizer calls it from ``clip_grad_norm`` and again from ``step``) and re-armed by
``step_done()``.  Averaging (1/world) is folded into the Adam kernel's ``grad_scale``.
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


def _all_gather_into(out, inp, group):
    """out [world * n, ...] <- concatenation of every rank's inp [n, ...] (async work handle)."""
    try:
        return dist.all_gather_into_tensor(out, inp, group=group, async_op=True)
    except (RuntimeError, NotImplementedError):     # backends without the flat form
        world = dist.get_world_size(group)
        return dist.all_gather(list(out.chunk(world, dim=0)), inp, group=group, async_op=True)


class BucketReducer:
    """Arena-slice all-reduce scheduler (works on any flat gradient tensor + [start, end)
    ranges, so it is testable on CPU with gloo)."""

    def __init__(self, flat_grad, process_group=None, use_side_stream=None):
        self.flat = flat_grad
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if use_side_stream is None:
            use_side_stream = flat_grad.is_cuda
        self.stream = torch.cuda.Stream() if use_side_stream else None
        self.pending = []
        self.enabled = True
        self.bytes_reduced = 0        # since the last finish(): payload of the launched collectives

    def _on_side_stream(self, launch):
        if self.stream is None:
            return launch()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            return launch()

    def reduce_range(self, start, end):
        if self.world == 1 or not self.enabled or end <= start:
            return
        buf = self.flat[start:end]
        self.pending.append(self._on_side_stream(
            lambda: dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)))
        self.bytes_reduced += buf.numel() * buf.element_size()

    def all_gather(self, out, inp):
        if self.world == 1:
            out.copy_(inp)
            return
        self.pending.append(self._on_side_stream(lambda: _all_gather_into(out, inp, self.pg)))
        self.bytes_reduced += out.numel() * out.element_size()

    def finish(self):
        """Block the compute stream (not the host) until every launched collective is done."""
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
        self.pg = process_group
        self.world = self.reducer.world
        self._arena = arena
        off = arena.offsets
        self._ranges = {'vocab': (0, off['position_embeddings.weight'][0]),
                        'embed': (off['position_embeddings.weight'][0], arena.embed_range[1]),
                        'heads': arena.head_range}
        for i, r in enumerate(arena.layer_ranges):
            self._ranges[('layer', i)] = r
        self.vocab_dense = True      # plan of the current step: does an MLM head feed the vocabulary matrix?
        self._live = 0               # encoder passes that still owe a backward
        self._live_streams = 0       # image-stream passes (ImageStreamFn) that still owe a backward: their gradients land
                                     # in the 'embed' range AFTER the encoder pass they feed has finished its own backward
        self._launched = set()
        self._tokens = []            # [(ids [n] int64, rows [n, d] bf16, n_max over ranks)]
        self._tokens_out = None
        self._finished = False
        self.exposed_events = None   # set to [] to record (start, end) events around finish()'s waits
        object.__setattr__(module, 'ddp_hook', self)    # plain attribute: as a registered submodule it would close a cycle
        if broadcast and self.world > 1:
            dist.broadcast(arena.master, src=0, group=process_group)
            for p in module.parameters():
                if getattr(p, '_m3p_arena', None) is None:
                    dist.broadcast(p.data, src=0, group=process_group)
            arena.mark_master_changed()

    def forward(self, mode, **kwargs):
        return self.module(mode, **kwargs)

    # ------------------------------------------------------------------ plan
    def plan_step(self, vocab_dense):
        """Called by the trainer at the start of a step type, identically on every rank:
        whether the step has an MLM head (a dense gradient for the vocabulary matrix)."""
        self.vocab_dense = bool(vocab_dense)

    def _launch(self, key):
        if self.world == 1 or not self.reducer.enabled or key in self._launched:
            return
        self._launched.add(key)
        self.reducer.reduce_range(*self._ranges[key])

    # ------------------------------------------------------------------ hooks (functional.py)
    @property
    def active(self):
        return self.world > 1

    def encoder_forward(self, n_tokens):
        """An encoder pass that will be differentiated.  Returns the largest token-row count of
        this pass over the ranks (ragged batches: the rows are padded to it for the all-gather);
        exchanged now, while the communication stream is idle."""
        self._live += 1
        if self.world == 1:
            return n_tokens
        t = torch.tensor([n_tokens], dtype=torch.int64, device=self._arena.device)
        if self.reducer.stream is not None:
            with torch.cuda.stream(self.reducer.stream):
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.pg)
                n_max = int(t.item())          # waits for the communication stream only
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.pg)
            n_max = int(t.item())
        return n_max

    def encoder_backward_begin(self):
        """-> True if this is the last pending encoder backward of the step."""
        last = self._live <= 1
        if last:
            if self.vocab_dense:
                self._launch('vocab')
            self._launch('heads')
        return last

    def encoder_backward_end(self):
        self._live = max(self._live - 1, 0)

    def mlm_head_done(self):
        if self.vocab_dense:
            self._launch('vocab')

    def layer_done(self, i, last=True):
        if last:
            self._launch(('layer', i))

    def embed_done(self, last=True, ids=None, rows=None, n_max=None):
        if ids is not None and self.world > 1:
            self._tokens.append((ids, rows, int(n_max)))
        if last:
            if self._live_streams == 0:      # else the image stream's backward still owes gradients inside this range
                self._launch('embed')
            self._exchange_tokens()
        self.encoder_backward_end()

    def stream_forward(self):
        """An image-stream pass (functional.ImageStreamFn) that will be differentiated: it feeds an encoder pass in
        layers-only mode, and its backward - the image projection / location / LayerNorm / language-table / refiner
        gradients, all inside the 'embed' range - runs after that encoder pass's."""
        self._live_streams += 1

    def stream_backward_end(self):
        self._live_streams = max(self._live_streams - 1, 0)
        if self._live_streams == 0 and self._live == 0:
            self._launch('embed')

    def _exchange_tokens(self):
        if self.world == 1 or not self.reducer.enabled or not self._tokens or self._tokens_out is not None:
            return
        dev = self._arena.device
        pad = int(self.module.pad_index)
        d = self._tokens[0][1].shape[1]
        n_tot = sum(nm for _, _, nm in self._tokens)
        ids_s = torch.full((n_tot,), pad, dtype=torch.int64, device=dev)
        rows_s = torch.empty((n_tot, d), dtype=torch.bfloat16, device=dev)
        o = 0
        for ids, rows, nm in self._tokens:
            n = ids.numel()
            ids_s[o:o + n] = ids.reshape(-1)
            rows_s[o:o + n] = rows
            if nm > n:
                rows_s[o + n:o + nm].zero_()
            o += nm
        ids_all = torch.empty((self.world * n_tot,), dtype=torch.int64, device=dev)
        rows_all = torch.empty((self.world * n_tot, d), dtype=torch.bfloat16, device=dev)
        self.reducer.all_gather(ids_all, ids_s)
        self.reducer.all_gather(rows_all, rows_s)
        self._tokens_out = (ids_all, rows_all, ids_s, rows_s)
        self._tokens = []

    def no_sync(self):
        """Context manager: skip the collectives on non-boundary micro-steps of gradient
        accumulation (the reference all-reduces on every micro-step, xtrainer.py:231-243).
        Token rows of such micro-steps are kept and exchanged at the boundary."""
        red = self.reducer

        class _Ctx:
            def __enter__(self_):
                red.enabled = False

            def __exit__(self_, *a):
                red.enabled = True
        return _Ctx()

    def finish(self):
        """Called before clip/Adam (from both): launch whatever the step's plan still owes (a
        step that ran no encoder backward, an encoder pass whose backward never came), wait,
        and apply the gathered token rows.  Idempotent until ``step_done()``."""
        if self.world == 1 or not self.reducer.enabled or self._finished:
            return
        if self.vocab_dense:
            self._launch('vocab')
        self._launch('heads')
        for i in reversed(range(len(self._arena.layer_ranges))):
            self._launch(('layer', i))
        self._launch('embed')
        self._exchange_tokens()
        ev0 = ev1 = None
        if self.exposed_events is not None and self._arena.device.type == 'cuda':
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        self.reducer.finish()
        if self._tokens_out is not None:
            from . import ops
            ids_all, rows_all = self._tokens_out[:2]
            ops.scatter_add_token_rows(rows_all, ids_all, self._arena.g('embeddings.weight'), self.module.pad_index)
            self._arena.touch('embeddings.weight')
            self._tokens_out = None
        if ev0 is not None:
            ev1.record()
            self.exposed_events.append((ev0, ev1, self.reducer.bytes_reduced))
        self.reducer.bytes_reduced = 0
        self._finished = True

    def step_done(self):
        """Re-arm for the next optimizer step (called by the fused optimizer / zero_grad)."""
        self._launched = set()
        self._finished = False
        self._live = 0
        self._live_streams = 0
        self._tokens = []
        self._tokens_out = None
