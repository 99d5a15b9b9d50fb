"""Data parallelism for the hot path: one process per MI355X, RCCL over xGMI through
``torch.distributed`` (backend "nccl" *is* RCCL on ROCm).

Replaces the reference's wiring at M3P/src/xtrainer.py:66-83 (Apex
``DistributedDataParallel(delay_allreduce=True)`` - one flat all-reduce of every gradient
*after* backward, fully exposed) and M3P/src/slurm.py:156-170 (process-group init).

Design.  The model's gradients live in ONE flat fp32 arena laid out in forward order, so a
"bucket" is an arena slice and nothing is copied, flattened or unflattened.  Bucket boundaries sit
on multiples of 512 elements (functional.Arena), so for 2 / 4 / 8 ranks every bucket splits into
equal 64-aligned shards and the exchange is the sharded-optimizer form (``mode = 'zero1'``,
numerically the all-reduce's result - SURVEY 8e):

  backward   ``reduce_scatter`` of a bucket the moment its last gradient kernel is enqueued, on a side
             stream, in this fixed, rank-independent order:
               1. ``vocab``   the tied vocabulary matrix + its bias (768 MB at V = 250 002): only its DENSE
                              part, the MLM-head weight gradient, final as soon as ``MLMHeadFn.backward``
                              returns - the first thing backward does - so the largest collective overlaps
                              with the whole encoder backward.  Steps without an MLM head never reduce it.
               2. ``heads``   pooler / relation / region heads, when the last encoder backward starts.
               3. ``layer i`` as soon as layer i's weight-gradient kernels are enqueued (28 MB at 768d).
               4. ``embed``   positions, embedding LayerNorm, image projection (+ refiner) after the assembly
                              backward (after the image stream's backward when the step has one).
               5. ``tokens``  the embedding-LOOKUP gradient, the only part of the vocabulary matrix produced
                              at the very end of backward, travels as what it is: <= B*T bf16 rows, all-gathered
                              with their ids and scatter-added in fp32 on every rank (50 MB per rank at B = 256
                              instead of a second pass over 768 MB).
  optimizer  the clip norm is the all-reduced sum of the ranks' shard norms (one 8-byte collective); Adam runs on
             this rank's shard of every bucket only - 1/world of the 9.5 GB the single-GPU step streams;
  forward    what the next forward needs of the other ranks' updates (round 4: half the bytes of round 3's fp32 master gather):
               * the **bf16 working copy** - Adam writes it together with the master for its shard - is ``all_gather``ed bucket
                 by bucket in FORWARD order on the side stream (0.56 GB per step instead of 1.12), the transposed copies
                 backward reads follow; the next step's forward waits per bucket (``params_ready``);
               * the parameters the kernels read in **fp32** (biases, LayerNorm weights, the position / location tables,
                 the ITM score vector - everything that is not a GEMM operand, under 1 M elements) travel as fp32: every rank
                 packs them, zeroes what it does not own, one ``all_reduce`` (each element has exactly one non-zero
                 contributor, so the sum is the owner's value bit for bit), and writes them back into its master.
             The fp32 master of the big MATRICES therefore stays sharded between checkpoints: a rank holds the current
             values of its own shards only (``master_partial``).  ``materialize_master()`` - a collective every rank must
             call, which ``Trainer.save_* / end_epoch`` do before their master-rank test - gathers the rest;
             ``TransformerModel.state_dict()`` refuses to hand out a partial master.  The Adam moments are never gathered: a
             checkpoint carries the saving rank's shards (the reference's reload restores ``num_updates`` only, and so does ours).

``mode = 'allreduce'`` (other world sizes, or ``M3P_DP_MODE=allreduce``) is round 2's protocol: fp32
``all_reduce`` per bucket, Adam replicated.  Half of the zero1 wire traffic moves from backward to the next
forward; the bytes on the wire are the same (ring all-reduce = reduce-scatter + all-gather).

A step may run more than one encoder pass (the CLCM objective runs ``jointfwd`` twice,
xtrainer.py:2379-2393): passes are counted in forward, and only the LAST backward of a step
launches layer / embed buckets - earlier ones just accumulate.  ``finish()`` launches
whatever the plan still owes, waits, applies the token rows; it is idempotent per step (the
optimizer calls it from ``clip_grad_norm`` and again from ``step``) and re-armed by
``step_done()``.  Averaging (1/world) is folded into the Adam kernel's ``grad_scale``.

No call in here blocks the host on the compute stream: the one host read of a step - the largest token-row count
of ragged batches, needed to size the all-gather - is requested in forward on the side stream and only read at the
end of backward (``_Pending``); batches the trainer declares uniform (``uniform_tokens``) skip it.

Compute units for RCCL next to the persistent 256-workgroup GEMMs: ``M3P_DP_RESERVE_CUS`` = r makes the GEMM grids leave r
CUs free (``m3p_set_persistent_grid``) and caps ``NCCL_MAX_NCHANNELS`` at r.  The default is r = 0: measured on one MI355X
with a 16-workgroup copy kernel on a side stream standing in for the collectives (tools/cu_reserve_ab.py,
profiles/r03_cu_reserve.txt), 2.7 GB of side traffic per step costs the step 2.6 ms whatever the grids leave free
(37.6 -> 40.3 ms at 256 workgroups, 41.2 -> 43.1 at 240), while the reservation itself costs 0.4 ms (8 CUs) to 3.6 ms
(16 CUs: 1968 tiles no longer deal out evenly) - the side kernel finds its slots at kernel boundaries either way.
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
    forced = os.environ.get('M3P_DP_FORCE') == '1' and 'MASTER_ADDR' in os.environ     # (tests: a one-rank world that still runs its collectives)
    if (world > 1 or forced) and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            # RCCL gets as many channels (one workgroup = one CU each) as the GEMM grids leave free
            if reserve_cus():
                os.environ.setdefault('NCCL_MAX_NCHANNELS', str(reserve_cus()))
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


def reserve_cus():
    """Compute units the persistent GEMM grids leave to RCCL when world > 1 (a multiple of 8: one per XCD)."""
    r = int(os.environ.get('M3P_DP_RESERVE_CUS', '0'))
    return max(0, min(r // 8 * 8, 64))


def _all_gather_into(out, inp, group):
    """out [world * n, ...] <- concatenation of every rank's inp [n, ...] (async work handle).  ``inp`` may be this
    rank's slot of ``out`` (the in-place form)."""
    try:
        return dist.all_gather_into_tensor(out, inp, group=group, async_op=True)
    except (RuntimeError, NotImplementedError):     # backends without the flat form (gloo)
        world = dist.get_world_size(group)
        return dist.all_gather(list(out.chunk(world, dim=0)), inp.clone(), group=group, async_op=True)


class _Done:
    def wait(self):
        pass


def _reduce_scatter_inplace(buf, rank, world, group):
    """Sum ``buf`` over the ranks; this rank's 1/world slice of it receives the result (the other slices keep
    whatever they held).  -> async work handle."""
    n = buf.numel() // world
    mine = buf[rank * n:(rank + 1) * n]
    if dist.get_backend(group) == 'nccl':
        return dist.reduce_scatter_tensor(mine, buf, op=dist.ReduceOp.SUM, group=group, async_op=True)
    # gloo has no reduce-scatter: all-reduce a copy and keep only this rank's slice - the other slices stay
    # un-reduced exactly as after RCCL's in-place reduce-scatter, so a test cannot lean on them
    tmp = buf.clone()
    dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group)
    mine.copy_(tmp[rank * n:(rank + 1) * n])
    return _Done()


class BucketReducer:
    """Arena-slice collective scheduler (works on any flat gradient tensor + [start, end)
    ranges, so it is testable on CPU with gloo)."""

    def __init__(self, flat_grad, process_group=None, use_side_stream=None):
        self.flat = flat_grad
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        # a world of one rank runs no collective - unless M3P_DP_FORCE=1 (tests: the RCCL branches on a one-GPU box)
        self.single = self.world == 1 and not (dist.is_initialized() and os.environ.get('M3P_DP_FORCE') == '1')
        if use_side_stream is None:
            use_side_stream = flat_grad.is_cuda
        self.stream = torch.cuda.Stream() if use_side_stream else None
        self.pending = []
        self.enabled = True
        self.bytes_reduced = 0        # since the last finish(): payload of the launched collectives
        self.timings = None           # set to [] to record (label, start event, end event, bytes, kind) per collective

    def _on_side_stream(self, launch, label=None, nbytes=0, kind='allreduce'):
        if self.stream is None:
            return launch()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            if self.timings is None:
                return launch()
            # (timing mode, bench.py: the collective is waited for on the side stream so that the end event brackets it)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            work = launch()
            work.wait()
            e1.record()
            self.timings.append((label, e0, e1, nbytes, kind))
            return work

    def reduce_range(self, start, end, label=None):
        if self.single or not self.enabled or end <= start:
            return
        buf = self.flat[start:end]
        nbytes = buf.numel() * buf.element_size()
        self.pending.append(self._on_side_stream(
            lambda: dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), label, nbytes, 'allreduce'))
        self.bytes_reduced += nbytes

    def reduce_scatter_range(self, start, end, label=None):
        """In-place reduce-scatter of flat[start:end]: rank r's 1/world slice ends up reduced."""
        if self.single or not self.enabled or end <= start:
            return
        assert (end - start) % self.world == 0
        buf = self.flat[start:end]
        nbytes = buf.numel() * buf.element_size()
        self.pending.append(self._on_side_stream(lambda: _reduce_scatter_inplace(buf, self.rank, self.world, self.pg),
                                                 label, nbytes, 'reduce_scatter'))
        self.bytes_reduced += nbytes

    def all_gather(self, out, inp, label=None):
        if self.single:
            out.copy_(inp)
            return
        nbytes = out.numel() * out.element_size()
        self.pending.append(self._on_side_stream(lambda: _all_gather_into(out, inp, self.pg), label, nbytes, 'all_gather'))
        self.bytes_reduced += nbytes

    def finish(self):
        """Block the compute stream (not the host) until every launched collective is done."""
        for w in self.pending:
            w.wait()
        self.pending = []
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _Pending:
    """A host value that a collective on the side stream is still producing (the largest token-row count of a ragged
    batch): requested in forward, read at the end of backward - by then the side stream has long passed it, so the
    read does not stall the host behind the compute stream."""

    def __init__(self, dp, n_local):
        dev = dp._arena.device
        red = dp.reducer
        if dev.type == 'cuda':
            host = torch.tensor([n_local], dtype=torch.int64).pin_memory()
            self._out = torch.empty(1, dtype=torch.int64).pin_memory()
            with (torch.cuda.stream(red.stream) if red.stream is not None else _null()):
                t = host.to(dev, non_blocking=True)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=dp.pg)
                self._out.copy_(t, non_blocking=True)
                self._ev = torch.cuda.Event()
                self._ev.record()
                self._keep = (t, host)
        else:
            t = torch.tensor([n_local], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=dp.pg)
            self._out, self._ev = t, None

    def value(self):
        if self._ev is not None:
            self._ev.synchronize()
            self._ev = self._keep = None
        return int(self._out[0])


class DataParallel(torch.nn.Module):
    """Wrapper with the surface the reference expects from (Apex) DDP: ``.module``,
    ``__call__(mode, **kw)`` forwarding, parameter broadcast from rank 0 at wrap time
    (xtrainer.py:68-83; used at :519-520, :814 and xevaluator.py:1532)."""

    def __init__(self, module, process_group=None, broadcast=True, mode=None):
        super().__init__()
        self.module = module
        arena = module.arena()
        self.reducer = BucketReducer(arena.grad, process_group)
        self.pg = process_group
        self.world = self.reducer.world
        self.single = self.reducer.single
        self.rank = self.reducer.rank
        self._arena = arena
        off = arena.offsets
        # buckets in FORWARD order (the order the parameter all-gathers of zero1 run in)
        self._ranges = {'vocab': (0, off['position_embeddings.weight'][0]),
                        'embed': (off['position_embeddings.weight'][0], arena.embed_range[1])}
        for i, r in enumerate(arena.layer_ranges):
            self._ranges[('layer', i)] = r
        self._ranges['heads'] = arena.head_range
        mode = mode or os.environ.get('M3P_DP_MODE') or 'zero1'
        assert mode in ('zero1', 'allreduce'), mode
        if mode == 'zero1' and any((e - s) % (64 * self.world) for s, e in self._ranges.values()):
            mode = 'allreduce'      # (a world size the 512-element bucket alignment does not divide into 64-aligned shards)
        self.mode = mode if not self.single else 'single'
        self.vocab_dense = True      # plan of the current step: does an MLM head feed the vocabulary matrix?
        self.uniform_tokens = False  # the trainer's promise that every rank's passes have the same token-row counts
        self._live = 0               # encoder passes that still owe a backward
        self._live_streams = 0       # image-stream passes (ImageStreamFn) that still owe a backward: their gradients land
                                     # in the 'embed' range AFTER the encoder pass they feed has finished its own backward
        self._launched = set()
        self._tokens = []            # [(ids [n] int64, rows [n, d] bf16, n_max over ranks: int or _Pending)]
        self._tokens_out = None
        self._finished = False
        self._param_events = {}      # zero1: bucket key -> event on the side stream (bf16 copy gathered); 'vectors', 'transposes'
        self.master_partial = False  # zero1: the fp32 master of the big matrices is current on its owner rank only
        self._vec = None             # (index int64 [n], own mask fp32 [n]) of the fp32-read parameters, built on first use
        self.exposed_events = None   # set to [] to record (start, end) events around finish()'s waits
        object.__setattr__(module, 'ddp_hook', self)    # plain attribute: as a registered submodule it would close a cycle
        if not self.single and arena.device.type == 'cuda' and os.environ.get('M3P_DP_TILE_QUEUE', '0') != '0':
            # the collectives' kernels share CUs with the persistent GEMMs: dynamic per-XCD tile queues let slowed-down CUs take
            # fewer tiles.  OFF by default since round 4: the queue instantiations cost 0.57 ms of an undisturbed step, and with
            # the exchange down to ~1.7 GB per step (bf16 parameter gather) the stand-in measurement of round 3 (static +8.4 ms,
            # queues +4.9 ms under 10.7 GB of side traffic) scales to about the same 1.4 ms either way - DESIGN.md section 5
            from . import ops
            ops.set_tile_queue(True)
        if not self.single and arena.device.type == 'cuda' and reserve_cus():
            from . import lib as L
            L.load().m3p_set_persistent_grid(L.num_cus() - reserve_cus())
        self._identity = None        # first-contact facts: gathered by identify() (a collective) only when somebody asks
        if broadcast and not self.single:
            dist.broadcast(arena.master, src=0, group=process_group)
            for p in module.parameters():
                if getattr(p, '_m3p_arena', None) is None:
                    dist.broadcast(p.data, src=0, group=process_group)
            arena.mark_master_changed()

    def identify(self):
        """COLLECTIVE on first use (every rank must call it: bench.py's multi-GPU half does; constructing DataParallel no longer
        does - ADVICE r5) - cached afterwards."""
        if self._identity is None:
            self._identity = self._identify()
        return self._identity

    def _identify(self):
        """First-contact facts about the process group (bench.py prints them under "comm"): the backend, how many ranks it
        has, and how many DISTINCT devices those ranks sit on - an all-gather of a 16-byte device tag (the uuid of the device,
        else host name + PCI bus id).  Eight ranks on one GPU (a launcher that did not set LOCAL_RANK, or HIP_VISIBLE_DEVICES
        pinned) reads `devices_seen: 1` here instead of as an unexplained 8x slowdown."""
        info = dict(backend=dist.get_backend(self.pg) if dist.is_initialized() else None, ranks=self.world, devices_seen=1)
        if self.single or not dist.is_initialized():
            return info
        import hashlib
        import socket
        dev = self._arena.device
        if dev.type == 'cuda':
            props = torch.cuda.get_device_properties(dev)
            tag = str(getattr(props, 'uuid', '')) or '%s/%s/%s' % (socket.gethostname(), getattr(props, 'pci_bus_id', dev.index),
                                                                 getattr(props, 'pci_device_id', ''))
            tag = socket.gethostname() + '/' + tag
        else:
            tag = '%s/cpu/%d' % (socket.gethostname(), os.getpid())
        mine = torch.frombuffer(bytearray(hashlib.md5(tag.encode()).digest()), dtype=torch.uint8).to(dev)
        every = torch.empty(self.world * 16, dtype=torch.uint8, device=dev)
        _all_gather_into(every, mine, self.pg).wait()
        info['devices_seen'] = len({bytes(r.tolist()) for r in every.view(self.world, 16).cpu()})
        return info

    def forward(self, mode, **kwargs):
        return self.module(mode, **kwargs)

    # ------------------------------------------------------------------ plan
    def plan_step(self, vocab_dense):
        """Called by the trainer at the start of a step type, identically on every rank:
        whether the step has an MLM head (a dense gradient for the vocabulary matrix)."""
        self.vocab_dense = bool(vocab_dense)

    def _launch(self, key):
        if self.single or not self.reducer.enabled or key in self._launched:
            return
        self._launched.add(key)
        st = self._arena.stale
        if st is not None and st[0] < self._ranges[key][1] and st[0] + st[1] > self._ranges[key][0]:
            # a lazily zeroed range (Arena.defer_vocab_zero) that no store has covered this step - a rank whose batch held no
            # masked word runs no MLM head - must be physically zero before it is summed with the other ranks' gradients
            self._arena.ensure_zero()
        if self.mode == 'zero1':
            self.reducer.reduce_scatter_range(*self._ranges[key], label=key)
        else:
            self.reducer.reduce_range(*self._ranges[key], label=key)

    # ------------------------------------------------------------------ shards (zero1)
    def shard_of(self, key):
        """This rank's [start, end) slice of bucket ``key``."""
        s, e = self._ranges[key]
        if self.mode != 'zero1':
            return s, e
        n = (e - s) // self.world
        return s + self.rank * n, s + (self.rank + 1) * n

    def owned(self, start, end):
        """The parts of [start, end) whose reduced gradient (and optimizer state) live on this rank: everything in
        all-reduce mode, the intersections with this rank's bucket shards in zero1 mode."""
        if self.mode != 'zero1':
            return [(start, end)]
        out = []
        for key in self._ranges:
            a, b = self.shard_of(key)
            a, b = max(a, start), min(b, end)
            if a < b:
                out.append((a, b))
        return out

    def all_reduce_scalar(self, t):
        """Sum a device scalar over the ranks on the current stream (the clip norm of sharded gradients)."""
        if self.mode == 'zero1':
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)

    def after_sharded_step(self, touched_ranges, keep=None):
        """zero1, called by the optimizer once Adam has run on this rank's shards: zero the rest of the touched gradient
        ranges (they hold this rank's un-reduced partials), exchange the fp32-read parameters (one packed all-reduce), then
        gather the updated bf16 working copy bucket by bucket in forward order on the side stream; the transposed copies
        come last.  The next forward waits per bucket (``params_ready``).  -> True if it took care of the bf16 copies.
        keep: [start, end) the optimizer left lazily un-zeroed (Arena.defer_vocab_zero: the next step's MLM head STORES over the
        whole vocabulary range, foreign shards included) - not zeroed here either."""
        if self.mode != 'zero1':
            return False
        ar = self._arena
        k0, k1 = keep if keep is not None else (0, 0)
        for s, e in touched_ranges:
            pos = s
            for a, b in self.owned(s, e) + [(e, e)]:
                if a > pos:
                    for z0, z1 in ((pos, min(a, k0)), (max(pos, k1), a)) if k1 > k0 else ((pos, a),):
                        if z1 > z0:
                            ar.grad[z0:z1].zero_()
                pos = max(pos, b)
        touched_keys = [k for k, (s, e) in self._ranges.items() if any(a < e and s < b for a, b in touched_ranges)]
        red = self.reducer
        cuda = ar.device.type == 'cuda'
        # (1) the fp32-read parameters: packed, masked to what this rank owns, summed over the ranks, written back
        idx, own = self._vector_index()
        if idx.numel():
            def _exchange():
                stage = ar.master.index_select(0, idx) * own
                if not self.single:
                    dist.all_reduce(stage, op=dist.ReduceOp.SUM, group=self.pg)
                ar.master.index_copy_(0, idx, stage)
            if cuda:
                ev0 = torch.cuda.Event()
                ev0.record(torch.cuda.current_stream())
                with torch.cuda.stream(red.stream):
                    red.stream.wait_event(ev0)
                    _exchange()
                    ev = torch.cuda.Event()
                    ev.record()
                self._param_events['vectors'] = ev
            else:
                _exchange()
        # (2) the bf16 working copy, bucket by bucket in forward order (Adam wrote this rank's shard of it with the master)
        for key in touched_keys:
            s, e = self._ranges[key]
            a, b = self.shard_of(key)
            red.all_gather(ar.w16[s:e], ar.w16[a:b], label=('params', key))
            work = red.pending.pop()
            if cuda:
                with torch.cuda.stream(red.stream):
                    work.wait()                                    # the side stream waits for the collective, the host does not
                    ev = torch.cuda.Event()
                    ev.record()
                self._param_events[key] = ev
            else:
                work.wait()
        if cuda:
            with torch.cuda.stream(red.stream):
                ar._transposes_stale = True
                ar.refresh_transposes()
                ev = torch.cuda.Event()
                ev.record()
            self._param_events['transposes'] = ev
        self.master_partial = True
        red.bytes_reduced = 0
        return True

    # Parameters the kernels consume through the bf16 working copy ONLY - every GEMM operand: the vocabulary matrix, the layers',
    # heads' and refiner's linear weights, the region projection - may keep their fp32 master sharded.  The kernels read in
    # fp32: biases and LayerNorm parameters (1-D), the small score / location projections, and the three tables named here.
    _MATRIX_MIN_ELEMS = 65536
    _FP32_TABLES = ('position_embeddings.weight', 'cross_lang_embeddings.weight',
                    'image_embeddings.image_location_embeddings.weight')

    def _is_sharded_matrix(self, name):
        o, cnt, shape = self._arena.offsets[name]
        return len(shape) == 2 and cnt >= self._MATRIX_MIN_ELEMS and name not in self._FP32_TABLES

    def _vector_index(self):
        if self._vec is None:
            ar = self._arena
            pieces, masks = [], []
            for name, (o, cnt, _) in ar.offsets.items():
                if self._is_sharded_matrix(name):
                    continue
                pieces.append(torch.arange(o, o + cnt, dtype=torch.int64))
                m = torch.zeros(cnt, dtype=torch.float32)
                for a, b in self.owned(o, o + cnt):
                    m[a - o:b - o] = 1.0
                masks.append(m)
            if pieces:
                self._vec = (torch.cat(pieces).to(ar.device), torch.cat(masks).to(ar.device))
            else:
                self._vec = (torch.zeros(0, dtype=torch.int64, device=ar.device), torch.zeros(0, device=ar.device))
        return self._vec

    def materialize_master(self):
        """Collective (every rank): gather the fp32 master shards of the big matrices so that every rank holds the full,
        current master - for checkpoints, ``state_dict()``, tests.  A no-op when nothing is partial."""
        if self.mode != 'zero1' or not self.master_partial:
            return
        self.params_ready(None)
        ar = self._arena
        for key, (s, e) in self._ranges.items():
            a, b = self.shard_of(key)
            if self.single:
                continue
            _all_gather_into(ar.master[s:e], ar.master[a:b], self.pg).wait()
        self.master_partial = False
        # (the bf16 copy is already current everywhere; the gather wrote the master in place: keep the cast version in step)
        if hasattr(ar, '_cast_version'):
            ar._cast_version = ar.master._version

    def params_ready(self, key=None):
        """Make the current stream wait until bucket ``key``'s parameters (None: all of them, and the transposed
        copies) are gathered and cast.  A stream-level wait; free when nothing is pending."""
        if not self._param_events:
            return
        ev = self._param_events.pop('vectors', None)      # the fp32-read parameters: every consumer needs them
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
        if key is None:
            for ev in self._param_events.values():
                torch.cuda.current_stream().wait_event(ev)
            self._param_events = {}
        else:
            ev = self._param_events.pop(key, None)
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)

    def full_reduced_grad(self):
        """The whole reduced gradient arena on every rank (tests / debugging): in zero1 mode the shards are gathered
        into a copy; in all-reduce mode it is the arena itself.  Call after ``finish()``."""
        g = self._arena.grad.clone()
        if self.mode == 'zero1':
            for key, (s, e) in self._ranges.items():
                if key in self._launched or key == 'vocab':      # (token rows are applied to a rank's own vocabulary shard only)
                    a, b = self.shard_of(key)
                    _all_gather_into(g[s:e], g[a:b].clone(), self.pg).wait()
        return g

    # ------------------------------------------------------------------ hooks (functional.py)
    @property
    def active(self):
        return not self.single

    def encoder_forward(self, n_tokens):
        """An encoder pass that will be differentiated.  Returns the largest token-row count of this pass over the
        ranks (ragged batches: the rows are padded to it for the all-gather) - as a value still in flight on the side
        stream (``_Pending``), read in ``embed_done``."""
        self._live += 1
        if self.single or self.uniform_tokens or n_tokens == 0:
            return n_tokens
        return _Pending(self, n_tokens)

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
        if ids is not None and not self.single:
            self._tokens.append((ids, rows, n_max))
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
        if self.single or not self.reducer.enabled or not self._tokens or self._tokens_out is not None:
            return
        dev = self._arena.device
        pad = int(self.module.pad_index)
        d = self._tokens[0][1].shape[1]
        sizes = [nm.value() if isinstance(nm, _Pending) else int(nm) for _, _, nm in self._tokens]
        n_tot = sum(sizes)
        # one message per rank: a row's id rides in eight extra bf16 columns behind its d values (the int64 in the first four;
        # a 16-byte tail keeps the rows 16-byte aligned) - ONE all-gather per step instead of two (round 5: each collective
        # launch costs the step boundary a 30-45 us hand-over between the compute and the exchange stream)
        pitch = d + 8
        msg = torch.empty((n_tot, pitch), dtype=torch.bfloat16, device=dev)
        ids_cols = msg[:, d:d + 4]
        o = 0
        for (ids, rows, _), nm in zip(self._tokens, sizes):
            n = ids.numel()
            msg[o:o + n, :d] = rows
            ids_cols[o:o + n] = ids.reshape(-1, 1).view(torch.bfloat16)
            if nm > n:
                msg[o + n:o + nm, :d].zero_()
                ids_cols[o + n:o + nm] = torch.full((1, 1), pad, dtype=torch.int64, device=dev).view(torch.bfloat16)
            o += nm
        msg_all = torch.empty((self.world * n_tot, pitch), dtype=torch.bfloat16, device=dev)
        ev = None
        if self.reducer.stream is not None and self.exposed_events is not None:
            # (bench: where the gradient buckets end and the token rows begin on the exchange stream - splits the exposed tail)
            ev = torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(self.reducer.stream):
                ev.record()
        self.reducer.all_gather(msg_all, msg, label='token rows')
        self._tokens_out = (msg_all, msg, d, ev)
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
        if self.single or not self.reducer.enabled or self._finished:
            return
        if self.vocab_dense:
            self._launch('vocab')
        self._launch('heads')
        for i in reversed(range(len(self._arena.layer_ranges))):
            self._launch(('layer', i))
        self._launch('embed')
        self._exchange_tokens()
        ev0 = ev1 = ev_tok = None
        if self.exposed_events is not None and self._arena.device.type == 'cuda':
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        self.reducer.finish()
        if self._tokens_out is not None:
            from . import ops
            msg_all, _, d_rows, ev_tok = self._tokens_out
            rows_all = msg_all[:, :d_rows]                                            # (strided view: pitch d + 8)
            ids_all = msg_all[:, d_rows:d_rows + 4].contiguous().view(torch.int64).view(-1)
            if self.mode == 'zero1' and self.world > 1:
                # sharded exchange: this rank steps (and norms) its own shard of the vocabulary gradient only, so of the N x
                # B x T gathered rows it needs those whose matrix row intersects that shard - the others become pad rows,
                # which the scatter kernel skips after reading their id (at 8 ranks: 1/8 of the atomics on average)
                o, cnt, shape = self._arena.offsets['embeddings.weight']
                a, b = self.shard_of('vocab')
                dd = shape[1]
                lo, hi = max(a - o, 0) // dd, -(-(min(b, o + cnt) - o) // dd)
                ids_all = torch.where((ids_all >= lo) & (ids_all < hi), ids_all,
                                      torch.full_like(ids_all, int(self.module.pad_index)))
            ops.scatter_add_token_rows(rows_all, ids_all, self._arena.g('embeddings.weight'), self.module.pad_index)
            self._arena.touch('embeddings.weight')
            self._tokens_out = None
        if ev0 is not None:
            ev1.record()
            self.exposed_events.append((ev0, ev1, self.reducer.bytes_reduced, ev_tok))
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
