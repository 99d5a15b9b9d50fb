"""Host-side composition of the HIP kernels into the encoder forward/backward and the MLM
head (autograd boundary).  Everything numeric happens in libm3p_hip.so; this file only
sequences launches, owns the flat parameter/gradient arenas and tells autograd where the
boundary is.

Gradient convention ("main-grad"): parameter gradients are accumulated by the kernels
straight into the fp32 gradient arena that ``param.grad`` views; the autograd Functions
return ``None`` for parameters.  That is what lets the data-parallel reducer all-reduce a
layer's gradients (one contiguous arena slice) the moment that layer's backward kernels
are enqueued, and lets clip+Adam run as flat streaming kernels.
"""
import math
import os
from collections import OrderedDict

import torch

from . import lib as L
from . import fp8 as fp8mod
from . import ops
from . import rng

BF16 = torch.bfloat16
ALIGN = 64
BUCKET_ALIGN = 64 * 8


def _round_up(n, a):
    return (n + a - 1) // a * a


# developer switch for A/B runs: 1 = gelu_fwd also leaves gelu'(u) in u's buffer and the dgrad uses EPI_MUL;
# 0 (default) = the dgrad epilogue recomputes gelu'(u) (EPI_DGELU).  -40 us per dgrad, +40 us per gelu pass.
_VOCAB_FULL_TILES = os.environ.get('M3P_VOCAB_FULL_TILES', '1') != '0'    # developer switch for A/B runs (Arena.V_pad)
_GELU_GRAD_IN_FWD = os.environ.get('M3P_GELU_GRAD_IN_FWD', '0') != '0'   # measured equal in the step (47.97 vs 47.97 ms): off
# round 4: gelu'(u) kept as ONE byte per element in the dU GEMM's fragment order (ops.gelu_fwd_gq / EPI_MULQ) instead of u itself;
# M3P_GELU_BYTE_GRAD=0 is the round-3 path (bf16 u + derivative table in the epilogue) for A/B runs
# 2 (default) = lin1's epilogue computes GELU and the byte itself (EPI_BIAS_GELUQ: no separate activation pass);
# 1 = bias epilogue + ops.gelu_fwd_gq pass
_GELU_BYTE_GRAD = int(os.environ.get('M3P_GELU_BYTE_GRAD', '2'))
# round 4: the vocabulary projection's epilogue leaves block-wise (max, sum exp) for the cross-entropy (EPI_BIAS_LSE); 0 = the
# round-3 path (a separate statistics pass over the logits) for A/B runs
_CE_FUSED_LSE = os.environ.get('M3P_CE_FUSED_LSE', '1') != '0'
# round 4: the vocabulary data gradient on the four-wave (tile, K-chunk) kernel instead of stream-K with atomics; 0 = round 3
_VOCAB_DGRAD_W4 = os.environ.get('M3P_VOCAB_DGRAD_W4', '1') != '0'

class Arena:
    """Flat storage behind a TransformerModel's hot parameters (see model/transformer.py)."""

    def __init__(self, model):
        named = model.hot_named_parameters()
        dev = model.embeddings.weight.device
        self.device = dev
        self.model = model
        self.names = list(named.keys())
        self.offsets = OrderedDict()
        off = 0
        # gradient buckets of the data-parallel reducer (distributed.py) start at these parameters; their boundaries sit
        # on multiples of BUCKET_ALIGN elements so that a bucket splits into equal, 64-element-aligned shards for 2 / 4 /
        # 8 ranks (reduce-scatter -> sharded Adam -> all-gather); the padding holds zeros in every arena
        L_ = model.n_layers
        bucket_starts = {'position_embeddings.weight', 'pooled_layer.dense.weight'} | \
            {'attentions.%d.q_lin.weight' % i for i in range(L_)}
        for n, p in named.items():
            if n in bucket_starts:
                off = _round_up(off, BUCKET_ALIGN)
            self.offsets[n] = (off, p.numel(), tuple(p.shape))
            off += _round_up(p.numel(), ALIGN)
        off = _round_up(off, BUCKET_ALIGN)
        self.total = off
        self.master = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.w16 = torch.zeros(off, dtype=BF16, device=dev)
        for n, p in named.items():
            o, cnt, shape = self.offsets[n]
            view = self.master[o:o + cnt].view(shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.grad[o:o + cnt].view(shape)
            p._m3p_arena = (self, n)
        self.params = named
        d, L_, V = model.dim, model.n_layers, model.n_words
        # pitch of the logits rows: whole 256-column tiles, so that the vocabulary projection of a 256-multiple of rows
        # runs on the eight-wave 256x256 kernel (it then reads rows V .. V_pad - 1 of "the matrix" - the bf16 bytes of
        # the parameters that follow embeddings.weight in the arena - and their logits land in the pad columns, which
        # the cross-entropy kernel zeroes again)
        self.V_pad = _round_up(V, 256)
        assert self.total >= self.V_pad * d, 'the arena ends before the padded vocabulary matrix does'
        # transposed bf16 copies for the data-gradient GEMMs
        self.wt = {}
        for i in range(L_):
            self.wt[('qkv', i)] = torch.zeros((d, 3 * d), dtype=BF16, device=dev)
            self.wt[('out', i)] = torch.zeros((d, d), dtype=BF16, device=dev)
            self.wt[('lin1', i)] = torch.zeros((d, 4 * d), dtype=BF16, device=dev)
            self.wt[('lin2', i)] = torch.zeros((4 * d, d), dtype=BF16, device=dev)
            if model.cross_attention_hot:
                self.wt[('xq', i)] = torch.zeros((d, d), dtype=BF16, device=dev)
                self.wt[('xkv', i)] = torch.zeros((d, 2 * d), dtype=BF16, device=dev)
                self.wt[('xout', i)] = torch.zeros((d, d), dtype=BF16, device=dev)
        # contiguous arena ranges used as gradient buckets (reverse-backward order) and by the optimizer
        self.head_range = (self.offsets['pooled_layer.dense.weight'][0], self.total)
        self.layer_ranges = []
        for i in range(L_):
            a = self.offsets['attentions.%d.q_lin.weight' % i][0]
            b = self.offsets['attentions.%d.q_lin.weight' % (i + 1)][0] if i + 1 < L_ else self.head_range[0]
            self.layer_ranges.append((a, b))
        self.embed_range = (0, self.layer_ranges[0][0] if L_ else self.head_range[0])
        self._cast_version = -1
        self._transposes_stale = True
        self._tdesc = None
        self.epoch = 0          # bumped whenever the weights change (load / cast / optimizer step): 8-bit copies key on it
        self.grads_known_zero = True
        self.touched = set()   # names of parameters that received gradient since the last zero_grad
        self.planned = set()   # the subset of `touched` that was only announced (plan()), not written yet
        # Lazy zero of the tied vocabulary matrix's gradient (round 5).  When a step's MLM head STORED its weight gradient over
        # grad[o : o + V_pad * d] (whole-tile sizes, first product into the zeroed matrix), the next step will almost always do
        # the same - so the fused Adam pass does not write 768 MB of zeros there only for them to be overwritten (0.77 GB of the
        # 9.5 GB it streams).  `stale` = (start, count) of a range that is LOGICALLY zero but physically still holds the last
        # gradient: the store path clears it for free, every other writer (g() hands out the views; an autograd hook covers
        # gradients autograd itself accumulates) and zero_grad() / readers go through ensure_zero() first.
        self.stale = None
        self.vocab_stored = False      # the last MLM-head backward took the store path
        self._vocab_range = None
        self._layer_names = [[n for n in self.names if n.startswith(('attentions.%d.' % i, 'layer_norm1.%d.' % i,
                                                                     'ffns.%d.' % i, 'layer_norm2.%d.' % i))]
                             for i in range(L_)]
        self._cross_names = [[n for n in self.names if n.startswith(('encoder_attn.%d.' % i, 'layer_norm15.%d.' % i))]
                             for i in range(L_)]

    def touch(self, *names):
        self.touched.update(names)
        self.planned.difference_update(names)
        self.grads_known_zero = False

    def plan(self, *names):
        """Mark parameters as part of this step's optimizer ranges BEFORE anything has written their gradient (data
        parallelism: a head that MAY run on some rank is stepped - and its range zeroed - on every rank).  Unlike touch() this
        does not count as a write: a kernel that wants to STORE over such a range (range_untouched) still may."""
        new = [n for n in names if n not in self.touched]
        self.planned.update(new)
        self.touched.update(new)
        self.grads_known_zero = False

    def range_untouched(self, start, count):
        """No parameter whose gradient overlaps grad[start : start + count] has received gradient since the last zero_grad
        (what a kernel that STORES into that range instead of accumulating needs to know)."""
        end = start + count
        return not any(o < end and o + cnt > start for n in self.touched if n not in self.planned for (o, cnt, _) in (self.offsets[n],))

    def touch_layer(self, i, cross=False):
        self.touched.update(self._layer_names[i])
        self.planned.difference_update(self._layer_names[i])
        if cross:
            self.touched.update(self._cross_names[i])
            self.planned.difference_update(self._cross_names[i])
        self.grads_known_zero = False

    # ---- views
    def w(self, name):
        """bf16 working copy of a parameter."""
        o, cnt, shape = self.offsets[name]
        return self.w16[o:o + cnt].view(shape)

    def g(self, name):
        o, cnt, shape = self.offsets[name]
        if self.stale is not None and o < self.stale[0] + self.stale[1] and o + cnt > self.stale[0]:
            self.ensure_zero()
        return self.grad[o:o + cnt].view(shape)

    def vocab_range(self):
        """(start, count) of what the vocabulary weight gradient's store covers: the matrix and its pad rows."""
        if self._vocab_range is None:
            self._vocab_range = (self.offsets['embeddings.weight'][0], self.V_pad * self.model.dim)
        return self._vocab_range

    def ensure_zero(self):
        """Make a lazily zeroed range physically zero (a writer that accumulates, or a reader, is about to touch it)."""
        if self.stale is not None:
            s0, cnt = self.stale
            self.stale = None
            self.grad[s0:s0 + cnt].zero_()

    def defer_vocab_zero(self):
        """Called by the optimizer instead of zeroing the vocabulary range.  Arms an autograd hook on the parameters inside the
        range once: gradients that autograd itself accumulates into param.grad (jointfwd(text_embed=...)) find zeros."""
        s0, cnt = self.vocab_range()
        self.stale = (s0, cnt)
        if not getattr(self, '_lazy_hooks', False):
            self._lazy_hooks = True
            for n, (o, c, _) in self.offsets.items():
                if o < s0 + cnt and o + c > s0:
                    def _hook(grad, _self=self):
                        _self.ensure_zero()
                        return grad
                    self.params[n].register_hook(_hook)

    def p(self, name):
        """The fp32 master of a parameter, for a kernel that reads it in fp32 (biases, LayerNorm weights, the position /
        language / location tables, the ITM score vector).  Under the sharded data-parallel exchange only the parameters that
        travel in the packed fp32 all-reduce are current on every rank - a big matrix's master is current on its owner rank
        only: handing THAT to a kernel would feed it stale weights on the other ranks, so it is refused here (the
        fp32-read set follows from the calls of this method, not from a list kept in step by hand)."""
        hook = getattr(self.model, 'ddp_hook', None)
        if hook is not None and getattr(hook, 'mode', None) == 'zero1' and hook._is_sharded_matrix(name):
            raise AssertionError('%s is a sharded matrix under zero1: its fp32 master is current on its owner rank only - read the '
                                 'bf16 working copy (Arena.w) or add it to distributed.DataParallel._FP32_TABLES' % name)
        return self.params[name]

    def qkv_w16(self, i):
        o = self.offsets['attentions.%d.q_lin.weight' % i][0]
        d = self.model.dim
        return self.w16[o:o + 3 * d * d].view(3 * d, d)

    def qkv_wgrad(self, i):
        o = self.offsets['attentions.%d.q_lin.weight' % i][0]
        d = self.model.dim
        return self.grad[o:o + 3 * d * d].view(3 * d, d)

    # encoder-attention sub-layer of layer i (cross_attention_hot): q | k | v adjacent like the self-attention
    def xattn(self, i, what, grad=False):
        """what: 'q' [d, d], 'kv' [2d, d] (weights; bf16 working copy or fp32 gradient), 'bq' [d], 'bkv' [2d] (fp32)."""
        d = self.model.dim
        if what in ('q', 'kv'):
            o = self.offsets['encoder_attn.%d.%s_lin.weight' % (i, 'q' if what == 'q' else 'k')][0]
            n = d if what == 'q' else 2 * d
            return (self.grad if grad else self.w16)[o:o + n * d].view(n, d)
        o = self.offsets['encoder_attn.%d.%s_lin.bias' % (i, 'q' if what == 'bq' else 'k')][0]
        return (self.grad if grad else self.master)[o:o + (d if what == 'bq' else 2 * d)]

    # AoA refiner layer i: the three input projections are adjacent in the arena like q/k/v of an encoder layer
    def ref_qkv_w16(self, i):
        o = self.offsets['refine_embeddings.layers.%d.self_attn.linears.0.weight' % i][0]
        d = self.model.dim
        return self.w16[o:o + 3 * d * d].view(3 * d, d)

    def ref_qkv_wgrad(self, i):
        o = self.offsets['refine_embeddings.layers.%d.self_attn.linears.0.weight' % i][0]
        d = self.model.dim
        return self.grad[o:o + 3 * d * d].view(3 * d, d)

    def ref_qkv_bias(self, i, grad=False):
        o = self.offsets['refine_embeddings.layers.%d.self_attn.linears.0.bias' % i][0]
        d = self.model.dim
        src = self.grad if grad else self.master
        return src[o:o + 3 * d]

    def qkv_bias(self, i, grad=False):
        o = self.offsets['attentions.%d.q_lin.bias' % i][0]
        d = self.model.dim
        src = self.grad if grad else self.master
        return src[o:o + 3 * d]

    # ---- freshness of the bf16 copies
    def mark_master_changed(self):
        """The WHOLE fp32 master was (re)written on this rank - load_state_dict, the data-parallel broadcast."""
        self._cast_version = -1
        self._transposes_stale = True
        self.epoch += 1
        hook = getattr(self.model, 'ddp_hook', None)
        if hook is not None:
            hook.master_partial = False

    def mark_updated_by_fused_optimizer(self):
        """Adam wrote master and w16 together through raw pointers: only transposes are stale."""
        self._cast_version = self.master._version
        self._transposes_stale = True
        self.epoch += 1

    def refresh(self, wait_params=True):
        """Bring the bf16 working copy and the transposed copies up to date with the fp32 master.  Under the sharded
        data-parallel exchange the copies are made on the side stream behind the parameter all-gathers; the current
        stream then only waits for them (all of them here; EncoderFn.forward waits bucket by bucket instead)."""
        hook = getattr(self.model, 'ddp_hook', None)
        if hook is not None and (wait_params or self._cast_version != self.master._version or self._transposes_stale):
            hook.params_ready(None)
        if self._cast_version != self.master._version:
            if hook is not None and getattr(hook, 'master_partial', False):
                # sharded data parallelism keeps the fp32 master of the big matrices current on its owner rank only: a cast
                # from it would overwrite the (current) bf16 copy with stale values
                raise RuntimeError('a parameter was modified in place while the fp32 master is sharded across the ranks: '
                                   'call DataParallel.materialize_master() on every rank first')
            L.check(L.load().m3p_cast_f32_bf16(self.master.data_ptr(), self.w16.data_ptr(), self.total, L.stream()),
                    'm3p_cast_f32_bf16')
            self._cast_version = self.master._version
            self._transposes_stale = True
        self.refresh_transposes()

    def refresh_transposes(self):
        if self._transposes_stale:
            if self._tdesc is None:
                rows = []
                mt = 1
                for i in range(self.model.n_layers):
                    for src, dst in ((self.qkv_w16(i), self.wt[('qkv', i)]),
                                     (self.w('attentions.%d.out_lin.weight' % i), self.wt[('out', i)]),
                                     (self.w('ffns.%d.lin1.weight' % i), self.wt[('lin1', i)]),
                                     (self.w('ffns.%d.lin2.weight' % i), self.wt[('lin2', i)])) + \
                                    (((self.xattn(i, 'q'), self.wt[('xq', i)]), (self.xattn(i, 'kv'), self.wt[('xkv', i)]),
                                      (self.w('encoder_attn.%d.out_lin.weight' % i), self.wt[('xout', i)]))
                                     if self.model.cross_attention_hot else ()):
                        r, c = src.shape
                        rows.append([src.data_ptr(), dst.data_ptr(), r, c, src.stride(0), dst.stride(0)])
                        mt = max(mt, ((r + 63) // 64) * ((c + 63) // 64))
                self._tdesc = (torch.tensor(rows, dtype=torch.int64, device=self.device), len(rows), mt) if rows else ()
            if self._tdesc:
                ops.transpose_batch(*self._tdesc)
            self._transposes_stale = False

    def zero_grad(self):
        """After this every gradient is PHYSICALLY zero (an explicit zero_grad is what callers outside the step protocol rely
        on - clip_grad_norm_ / p.grad.add_ on model.parameters(), a wrapper attached after single-GPU steps: ADVICE r5); the
        lazily zeroed vocabulary range is only ever left behind by the fused optimizer step (after_fused_step)."""
        if not self.grads_known_zero:
            self.grad.zero_()
            self.grads_known_zero = True
            self.stale = None
        self.ensure_zero()
        self.vocab_stored = False
        self.touched.clear()
        self.planned.clear()
        if self.model.ddp_hook is not None:
            self.model.ddp_hook.step_done()

    def after_fused_step(self, copies_scheduled=False):
        """The fused Adam kernel updated master + w16 and zeroed every touched gradient range.  copies_scheduled: the
        data-parallel hook already queued the bf16 cast and the transposes behind its parameter all-gathers."""
        self.mark_updated_by_fused_optimizer()
        if copies_scheduled:
            self._transposes_stale = False
        self.touched.clear()
        self.planned.clear()
        self.grads_known_zero = True
        self.vocab_stored = False
        if self.model.ddp_hook is not None:
            self.model.ddp_hook.step_done()


def _site(kind, layer=0):
    return {'img': 0, 'emb': 1}.get(kind, 8 + 4 * layer + {'attn_p': 0, 'attn_out': 1, 'ffn': 2}.get(kind, 3))


_DEC_SITE0 = 1 << 21   # dropout sites of the causal stream: _DEC_SITE0 + 8 * layer + {0 self-attention probabilities,
                        # 1 self-attention output, 2 encoder-attention probabilities, 3 encoder-attention output, 4 FFN}, + 7 = embeddings


class ArenaCrossWeights:
    """The encoder-attention weights as the decoder path wants them (q, kv, out, bq, bkv, bo per layer), read from the arena."""

    def __init__(self, ar):
        n = ar.model.n_layers
        self.q = [ar.xattn(i, 'q') for i in range(n)]
        self.kv = [ar.xattn(i, 'kv') for i in range(n)]
        self.out = [ar.w('encoder_attn.%d.out_lin.weight' % i) for i in range(n)]
        self.bq = [ar.xattn(i, 'bq') for i in range(n)]
        self.bkv = [ar.xattn(i, 'bkv') for i in range(n)]
        self.bo = [ar.p('encoder_attn.%d.out_lin.bias' % i) for i in range(n)]


_REF_SITE0 = 1 << 20   # dropout sites of the refiner: _REF_SITE0 + 8 * layer + {0 attention probabilities, 1 AoA
                        # input, 2 attention sublayer, 3 TransformerFFN's own, 4 FFN sublayer}


def _transposed(w):
    """bf16 [n, k] -> fresh [k, n] (the refiner's weights are small: transposed per use instead of kept)."""
    wt = torch.empty((w.shape[1], w.shape[0]), dtype=BF16, device=w.device)
    ops.transpose_bf16(w, wt)
    return wt


def refiner_fwd(model, x, keylen, B, R, seed_step, p):
    """AoA_Refiner_Core.forward (transformer.py:410-422) on the image rows x (bf16 [B*R, d], rows b*R + r):
    per layer  x = x + drop(AoA(LN_a(x)));  x = x + drop(FFN(LN_b(x)))  (pre-norm SublayerConnection :381-394),
    then the final LayerNorm.  AoA (MultiHeadedDotAttention with do_aoa, :327-371): q/k/v projections of the
    normed rows, softmax(q k^T / sqrt(d_k), key mask) with dropout, GLU(Linear(dropout(cat[attended, normed]))).
    p: the refiner's dropout rate - the reference hard-codes 0.1 through its constructor defaults (:288,:411,:662),
    independent of params.dropout.  Returns (rows, saved)."""
    ar = model.arena()
    d, H = model.dim, model.n_heads
    dh = d // H
    M = B * R
    qscale = 1.0 / math.sqrt(dh)
    seed = lambda i, k: rng.stream_seed(model.base_seed, seed_step, _REF_SITE0 + 8 * i + k)   # noqa: E731
    layers = []
    for i in range(model.n_refine_layers):
        pre = 'refine_embeddings.layers.%d.' % i
        xn, mean_a, rstd_a = ops.layernorm_fwd(x, ar.p(pre + 'sublayer.0.norm.weight'), ar.p(pre + 'sublayer.0.norm.bias'))
        qkv = ops.gemm_nt(xn, ar.ref_qkv_w16(i), L.EPI_BIAS, bias=ar.ref_qkv_bias(i), scale_cols=d, scale=qscale)
        ctxt, lse, kmask = ops.attn_fwd(qkv, keylen, B, R, H, dh, seed=seed(i, 0), p_drop=p, want_mask=True)
        cat = torch.empty((M, 2 * d), dtype=BF16, device=x.device)
        ops.dropout_rows(ctxt, p, seed(i, 1), out=cat[:, :d], rng_ld=2 * d, rng_col0=0)
        ops.dropout_rows(xn, p, seed(i, 1), out=cat[:, d:], rng_ld=2 * d, rng_col0=d)
        ab = ops.gemm_nt(cat, ar.w(pre + 'self_attn.aoa_layer.0.weight'), L.EPI_BIAS,
                         bias=ar.p(pre + 'self_attn.aoa_layer.0.bias'))
        x1 = ops.dropout_rows(ops.glu_fwd(ab), p, seed(i, 2), res=x)
        xn2, mean_b, rstd_b = ops.layernorm_fwd(x1, ar.p(pre + 'sublayer.1.norm.weight'), ar.p(pre + 'sublayer.1.norm.bias'))
        u = torch.empty((M, 4 * d), dtype=BF16, device=x.device)
        hact = ops.gemm_nt(xn2, ar.w(pre + 'feed_forward.lin1.weight'), L.EPI_BIAS_GELU,
                           bias=ar.p(pre + 'feed_forward.lin1.bias'), out2=u)
        t = ops.gemm_nt(hact, ar.w(pre + 'feed_forward.lin2.weight'), L.EPI_BIAS, bias=ar.p(pre + 'feed_forward.lin2.bias'))
        if p > 0:
            ops.dropout_rows(t, p, seed(i, 3), out=t)            # TransformerFFN.forward's dropout (:226) ...
        x2 = ops.dropout_rows(t, p, seed(i, 4), res=x1)          # ... then the sublayer's (:394)
        layers.append((x, mean_a, rstd_a, xn, qkv, ctxt, lse, kmask, cat, ab, x1, mean_b, rstd_b, xn2, u, hact))
        x = x2
    out, mean_f, rstd_f = ops.layernorm_fwd(x, ar.p('refine_embeddings.norm.weight'), ar.p('refine_embeddings.norm.bias'))
    return out, (layers, x, mean_f, rstd_f)


def refiner_bwd(model, dout, saved, keylen, B, R, seed_step, p):
    """Gradient of refiner_fwd's input; parameter gradients go to the arena."""
    ar = model.arena()
    d, H = model.dim, model.n_heads
    dh = d // H
    seed = lambda i, k: rng.stream_seed(model.base_seed, seed_step, _REF_SITE0 + 8 * i + k)   # noqa: E731
    layers, x_last, mean_f, rstd_f = saved
    dx, _ = ops.layernorm_bwd(dout, None, x_last, ar.p('refine_embeddings.norm.weight'), mean_f, rstd_f, None,
                              ar.g('refine_embeddings.norm.weight'), ar.g('refine_embeddings.norm.bias'))
    for i in reversed(range(model.n_refine_layers)):
        pre = 'refine_embeddings.layers.%d.' % i
        (x, mean_a, rstd_a, xn, qkv, ctxt, lse, kmask, cat, ab, x1, mean_b, rstd_b, xn2, u, hact) = layers[i]
        layers[i] = None
        # ---- x2 = x1 + drop4(drop3(hact W2^T + b2))
        dt = ops.dropout_rows(dx, p, seed(i, 4))
        if p > 0:
            ops.dropout_rows(dt, p, seed(i, 3), out=dt)
        ops.colsum(dt, d, ar.g(pre + 'feed_forward.lin2.bias'))
        ops.gemm_wgrad(dt, hact, ar.g(pre + 'feed_forward.lin2.weight'))
        du = ops.gemm_nt(dt, _transposed(ar.w(pre + 'feed_forward.lin2.weight')), L.EPI_DGELU, aux=u,
                         colsum=ar.g(pre + 'feed_forward.lin1.bias'))
        ops.gemm_wgrad(du, xn2, ar.g(pre + 'feed_forward.lin1.weight'))
        dxn2 = ops.gemm_nt(du, _transposed(ar.w(pre + 'feed_forward.lin1.weight')), L.EPI_NONE)
        dln, _ = ops.layernorm_bwd(dxn2, None, x1, ar.p(pre + 'sublayer.1.norm.weight'), mean_b, rstd_b, None,
                                   ar.g(pre + 'sublayer.1.norm.weight'), ar.g(pre + 'sublayer.1.norm.bias'))
        dx1 = ops.dropout_rows(dln, 0.0, 0, res=dx)               # residual: dx1 = dx + dLN
        del dt, du, dxn2, dln, u, hact, xn2
        # ---- x1 = x + drop2(GLU(cat W_aoa^T + b)),  cat = drop1([ctx | xn])
        dy = ops.dropout_rows(dx1, p, seed(i, 2))
        dab = ops.glu_bwd(ab, dy)
        ops.colsum(dab, 2 * d, ar.g(pre + 'self_attn.aoa_layer.0.bias'))
        ops.gemm_wgrad(dab, cat, ar.g(pre + 'self_attn.aoa_layer.0.weight'))
        dcat = ops.gemm_nt(dab, _transposed(ar.w(pre + 'self_attn.aoa_layer.0.weight')), L.EPI_NONE)
        dctx = ops.dropout_rows(dcat[:, :d], p, seed(i, 1), rng_ld=2 * d, rng_col0=0)
        dxn_cat = ops.dropout_rows(dcat[:, d:], p, seed(i, 1), rng_ld=2 * d, rng_col0=d)
        dqkv = ops.attn_bwd(qkv, keylen, ctxt, dctx, lse, B, R, H, dh, dbias_qkv=ar.ref_qkv_bias(i, grad=True),
                            seed=seed(i, 0), p_drop=p, keepmask=kmask)
        ops.gemm_wgrad(dqkv, xn, ar.ref_qkv_wgrad(i))
        dxn = ops.gemm_nt(dqkv, _transposed(ar.ref_qkv_w16(i)), L.EPI_RES, aux=dxn_cat)
        dln, _ = ops.layernorm_bwd(dxn, None, x, ar.p(pre + 'sublayer.0.norm.weight'), mean_a, rstd_a, None,
                                   ar.g(pre + 'sublayer.0.norm.weight'), ar.g(pre + 'sublayer.0.norm.bias'))
        dx = ops.dropout_rows(dln, 0.0, 0, res=dx1)
        del dy, dab, dcat, dctx, dxn_cat, dqkv, dxn, dln, dx1
    ar.touch(*[n for n in ar.names if n.startswith('refine_embeddings.')])
    return dx


class GradSink:
    """Where the heads of one encoder pass leave the gradient of its output.  The reference slices / transposes the
    (S, B, d) output before every head (xtrainer.py:2287-2289, :2357), and autograd's backward of each of those views is
    a zero-filled copy of the whole 64-MB tensor plus an add per extra head; here every head scatter-adds its few rows
    into ONE zeroed row buffer (allocated by the first head that arrives) and returns no gradient for its input -
    EncoderFn.backward takes the buffer.  Heads outside the protocol still work: their gradient arrives through
    autograd and is added to the buffer's."""

    def __init__(self):
        self.buf = None

    def rows(self, like):
        if self.buf is None:
            self.buf = torch.zeros_like(like)
        assert self.buf.shape == like.shape
        return self.buf

    def take(self):
        buf, self.buf = self.buf, None
        return buf


def _sink_of(tensor, base):
    """The GradSink of the encoder pass ``tensor`` is a view of (None if it is not one, or if ``base`` - the [rows, d]
    buffer the head indexes - is not exactly that pass's output)."""
    root = tensor._base if tensor._is_view() else tensor
    sink = getattr(root, '_m3p_sink', None) if root is not None else None
    if sink is None or base is None or not torch.is_grad_enabled():
        return None
    if root.dim() != 2 or not root.is_contiguous() or root.storage_offset() != 0 or tuple(root.shape) != tuple(base.shape):
        return None
    return sink


def first_rows_sink(model, tensor):
    """For predict(is_relation=True) on ``tensor`` (B, S, d): (sink, base, rows) with rows = the row numbers of
    tensor[:, 0] inside the encoder pass's [M, d] output, when tensor is a view of one; (None, None, None) otherwise."""
    d = model.dim
    if tensor.dim() != 3 or tensor.dtype != BF16 or not tensor._is_view():
        return None, None, None
    st, soff = tensor.stride(), tensor.storage_offset()
    if st[2] != 1 or st[0] % d or soff % d:
        return None, None, None
    with torch.no_grad():
        base = torch.as_strided(tensor, (tensor.untyped_storage().nbytes() // 2 // d, d), (d, 1), 0)
    sink = _sink_of(tensor, base)
    if sink is None:
        return None, None, None
    cache = model.__dict__.setdefault('_first_rows_cache', {})
    key = (tensor.shape[0], st[0], soff, tensor.device)
    rows = cache.get(key)
    if rows is None:
        rows = ((soff + torch.arange(tensor.shape[0], device=tensor.device, dtype=torch.int64) * st[0]) // d).to(torch.int32)
        cache.clear()
        cache[key] = rows
    return sink, base, rows


N_ENC_ARGS = 16     # positional arguments of EncoderFn.forward (backward returns one None per argument)


class EncoderFn(torch.autograd.Function):
    """Embedding assembly + n_layers post-LN transformer layers
    (TransformerModel.jointfwd, M3P/src/model/transformer.py:901-958; with x_img=None the
    text stream of crossfwd, :1050-1102).  ``anchor`` is any parameter: it only makes the
    output require grad; parameter gradients are written to the arena (module docstring)."""

    @staticmethod
    def forward(ctx, anchor, model, x, lengths, x_img, lengths_img, image_loc, p_drop, p_attn, seed_step, p_refine=None,
                track=False, text_embed=None, langs=None, h0=None, positions=None):
        ar = model.arena()
        # sharded data parallelism: the parameters of this step arrive bucket by bucket (in forward order) on the side
        # stream; wait for what the embedding stage reads now and for each layer in front of its first GEMM
        ready = model.ddp_hook.params_ready if model.ddp_hook is not None else (lambda key: None)
        ar.refresh(wait_params=False)
        ready('vocab')
        ready('embed')
        dev = ar.device
        d, H, nL = model.dim, model.n_heads, model.n_layers
        dh = d // H
        seed = lambda kind, i=0: rng.stream_seed(model.base_seed, seed_step, _site(kind, i))   # noqa: E731
        if h0 is not None:
            # the layers alone on rows assembled elsewhere (the image-only stream, ImageStreamFn): h0 bf16 [B*S, d],
            # rows b*S + s, already masked; its gradient goes back to autograd
            assert x is None and x_img is None and text_embed is None and langs is None and positions is None
            B = lengths.shape[0]
            S = h0.shape[0] // B
            T, R, M = S, 0, B * S
            totlen, rowmask = ops.seq_masks(lengths.to(device=dev, dtype=torch.int64).contiguous(), None, B, S)
            h = h0.detach().to(BF16).contiguous()
            ximg16 = loc = emb_saved = img_rows = img_saved = ref_saved = keylen_img = tok = None
        else:
            h = None
            T, B = x.shape
            R = 0 if x_img is None else x_img.shape[0]
            S = R + T
            M = B * S
            x = x.to(dev).contiguous()
        if h is None:
            # text_embed (transformer.py:910-913, the FreeLB steps' perturbed embeddings, (B, T, d)): the assembly kernel
            # gathers "token" b*T + t from the rows of text_embed instead of id x[t, b] from the vocabulary matrix
            table, tok = ar.w('embeddings.weight'), x
            if text_embed is not None:
                assert tuple(text_embed.shape) == (B, T, d), (tuple(text_embed.shape), (B, T, d))
                table = text_embed.detach().to(device=dev, dtype=BF16).contiguous().view(B * T, d)
                tok = (torch.arange(B, device=dev, dtype=torch.int64) * T)[None, :] + \
                    torch.arange(T, device=dev, dtype=torch.int64)[:, None]
                tok = tok.contiguous()
            if langs is not None or positions is not None:
                # language embeddings of the text stream (transformer.py:1059-1060): the assembly kernel again gathers "token"
                # b*T + t, now from the rows  Emb[x] + Lang[langs]  built here (one extra bf16 rounding of the sum); backward
                # gets the rows' gradients back and scatters them into both tables.  Explicit positions (TLM batches restart
                # them at the second sentence, utils.py:324 concat_batches; transformer.py:1057-1058) enter the same rows as
                # P[pos] - P[t]: the kernel adds P[t] back
                assert text_embed is None and R == 0
                rows = table[x.t()].float()
                if langs is not None:
                    langs = langs.to(dev).contiguous()
                    rows = rows + ar.p('cross_lang_embeddings.weight')[langs.t()]
                if positions is not None:
                    positions = positions.to(dev).contiguous()
                    assert positions.size() == (T, B)
                    ptab = ar.p('position_embeddings.weight')
                    rows = rows + ptab[positions.t()] - ptab[:T][None, :, :]
                table = rows.to(BF16).reshape(B * T, d).contiguous()
                tok = ((torch.arange(B, device=dev, dtype=torch.int64) * T)[None, :] +
                       torch.arange(T, device=dev, dtype=torch.int64)[:, None]).contiguous()
            # prefix validity of [regions | words] (get_masks on lengths + lengths_img, transformer.py:917-919): one launch
            totlen, rowmask = ops.seq_masks(lengths.to(device=dev, dtype=torch.int64).contiguous(),
                                            None if R == 0 else lengths_img.to(device=dev, dtype=torch.int64).contiguous(), B, S)

            ximg16 = img_proj = loc = None
            if R > 0:
                xi = x_img.detach().to(dev)
                if xi.dtype == torch.float32 and xi.stride(2) == 1 and xi.stride(0) % 4 == 0 and xi.stride(1) % 4 == 0:
                    ximg16 = ops.cast_rows_bf16(xi)         # reads the collate's (n, R, 2048) layout through the (R, n) view
                else:
                    ximg16 = ops.cast_bf16(xi.contiguous().view(R * B, 2048))
                loc = image_loc.contiguous().float()
                img_proj = ops.gemm_nt(ximg16, ar.w('image_embeddings.image_embeddings.weight'), L.EPI_BIAS,
                                       bias=ar.p('image_embeddings.image_embeddings.bias'))
            # refine_image (transformer.py:905-906): the image rows take a detour through the AoA refiner
            img_rows = img_saved = ref_saved = keylen_img = None
            if p_refine is not None and R > 0:
                rows, img_saved = ops.embed_image_rows_fwd(
                    img_proj, loc, ar.p('image_embeddings.image_location_embeddings.weight'),
                    ar.p('image_embeddings.image_location_embeddings.bias'), ar.p('image_embeddings.LayerNorm.weight'),
                    ar.p('image_embeddings.LayerNorm.bias'), B, R, d, seed_img=seed('img'), p_drop=p_drop)
                keylen_img = lengths_img.to(device=dev, dtype=torch.int32).contiguous()
                img_rows, ref_saved = refiner_fwd(model, rows, keylen_img, B, R, seed_step, p_refine)
            h, emb_saved = ops.embed_assemble_fwd(
                tok, table, ar.p('position_embeddings.weight'), img_proj, loc,
                ar.p('image_embeddings.image_location_embeddings.weight'),
                ar.p('image_embeddings.image_location_embeddings.bias'),
                ar.p('image_embeddings.LayerNorm.weight'), ar.p('image_embeddings.LayerNorm.bias'),
                ar.p('layer_norm_emb.weight'), ar.p('layer_norm_emb.bias'), totlen, B, T, R, d,
                seed_img=seed('img'), seed_emb=seed('emb'), p_drop=p_drop, img_rows=img_rows, img_saved=img_saved)

        saved_layers = []
        qscale = 1.0 / math.sqrt(dh)
        # fp8 GEMMs (BASELINE configs[3]; m3p_amd/fp8.py): the layer projections run on 8-bit operands
        st8 = model.fp8_state() if model.fp8 else None
        if st8 is not None:
            assert M % 256 == 0, 'the fp8 GEMM takes whole 256-row tiles (B * S %% 256 == 0): M = %d' % M
            if model.training:
                st8.roll()
            ready(None)                 # (the batched weight quantisation reads every layer's copy)
            st8.quant_weights(ar)

        def lin(xin, i, xsite, wsite, w16, epi, pre8=None, **kw):
            if st8 is None or wsite not in fp8mod.FWD_SITES:
                return ops.gemm_nt(xin, w16, epi, **kw)
            x8, dxs = pre8 if pre8 is not None else st8.quant(xin, i, xsite)
            w8, _, dws = st8.weights[(i, wsite)]
            return ops.gemm_nt_fp8(x8, w8, epi, descale_a=dxs, descale_b=dws, **kw)

        for i in range(nL):
            a, f = 'attentions.%d.' % i, 'ffns.%d.' % i
            ready(('layer', i))
            qkv = lin(h, i, 'x', 'wqkv', ar.qkv_w16(i), L.EPI_BIAS, bias=ar.qkv_bias(i), scale_cols=d, scale=qscale)
            ctxt, lse, kmask = ops.attn_fwd(qkv, totlen, B, S, H, dh, seed=seed('attn_p', i), p_drop=p_attn,
                                            want_mask=True)
            lse = (lse, kmask)      # the dropout keep bits travel with the log-sum-exp to backward
            pre1 = lin(ctxt, i, 'ctx', 'wout', ar.w(a + 'out_lin.weight'), L.EPI_BIAS_DROP_RES, bias=ar.p(a + 'out_lin.bias'),
                       aux=h, seed=seed('attn_out', i), p_drop=p_drop)
            x1, mean1, rstd1 = ops.layernorm_fwd(pre1, ar.p('layer_norm1.%d.weight' % i), ar.p('layer_norm1.%d.bias' % i))
            hact8 = None
            u_holds_grad = False     # does the pass below leave gelu'(u) in u's buffer? (backward then only multiplies)
            if (st8 is None or 'w1' not in fp8mod.FWD_SITES) and not _GELU_GRAD_IN_FWD and _GELU_BYTE_GRAD == 2 and track \
                    and ops.gq_eligible(M, 4 * d):
                # lin1 + GELU in ONE launch: the epilogue writes h and, for backward, gelu'(u) as one byte per element in the
                # dU GEMM's own fragment order (EPI_MULQ decodes it with one fma); u is never stored
                u = torch.empty((M * 4 * d,), dtype=torch.uint8, device=dev)
                o8 = {}
                if st8 is not None and 'w2' in fp8mod.FWD_SITES:
                    # fp8 (round 6): the same epilogue leaves the e4m3 copy of h that the 8-bit lin2 product reads - no
                    # quantisation pass, no GELU pass (a site's first use has no scale yet: bf16 once, its maximum measured)
                    kh = st8.index(i, 'hact')
                    if st8.seen[kh]:
                        o8 = dict(out8=torch.empty((M, 4 * d), dtype=torch.uint8, device=dev), scale8=st8.scale[kh:kh + 1],
                                  amax8=st8.amax[kh:kh + 1])
                hact = ops.gemm_nt(x1, ar.w(f + 'lin1.weight'), L.EPI_BIAS_GELUQ, bias=ar.p(f + 'lin1.bias'), out2=u, **o8)
                if o8:
                    hact8 = (o8['out8'], st8.descale[kh:kh + 1])
                elif st8 is not None and 'w2' in fp8mod.FWD_SITES:
                    st8._first_use(hact, kh)
                u_holds_grad = 'q'
            elif M >= 1024 or st8 is not None:
                # persistent GEMM: bias in the epilogue, GELU as its own HBM-speed pass
                # The same pass leaves gelu'(u) in u's buffer: the backward dgrad then only multiplies
                # (EPI_MUL, runs on the four-wave GEMM) instead of evaluating erf/exp in its epilogue.
                u = lin(x1, i, 'x1', 'w1', ar.w(f + 'lin1.weight'), L.EPI_BIAS, bias=ar.p(f + 'lin1.bias'))
                if st8 is not None and 'w2' in fp8mod.FWD_SITES and 'w2' not in fp8mod.BWD_SITES and not _GELU_GRAD_IN_FWD:
                    hact, hact8 = st8.gelu_quant(u, i)           # GELU and the 8-bit copy for lin2 in one pass
                elif st8 is None and not _GELU_GRAD_IN_FWD and _GELU_BYTE_GRAD and track and ops.gq_eligible(M, 4 * d):
                    # what backward needs of u is gelu'(u): one byte per element in the dU GEMM's own fragment order
                    # (EPI_MULQ decodes it with one fma - no derivative table, no aux trip through LDS); u is dropped here
                    hact, u = ops.gelu_fwd_gq(u)
                    u_holds_grad = 'q'
                else:
                    u_holds_grad = _GELU_GRAD_IN_FWD or (st8 is not None and 'w2' in fp8mod.BWD_SITES)
                    hact = ops.gelu_fwd(u, grad_inplace=u_holds_grad)
            else:
                u = torch.empty((M, 4 * d), dtype=BF16, device=dev)
                hact = ops.gemm_nt(x1, ar.w(f + 'lin1.weight'), L.EPI_BIAS_GELU, bias=ar.p(f + 'lin1.bias'), out2=u)
            pre2 = lin(hact, i, 'hact', 'w2', ar.w(f + 'lin2.weight'), L.EPI_BIAS_DROP_RES, bias=ar.p(f + 'lin2.bias'),
                       aux=x1, seed=seed('ffn', i), p_drop=p_drop, pre8=hact8)
            h_next, mean2, rstd2 = ops.layernorm_fwd(pre2, ar.p('layer_norm2.%d.weight' % i),
                                                     ar.p('layer_norm2.%d.bias' % i), rowmask)
            if track:      # (inference keeps nothing: retrieval evaluation runs thousands of sequences per call)
                saved_layers.append((h, qkv, ctxt, lse, pre1, mean1, rstd1, x1, u, hact, pre2, mean2, rstd2))
            h = h_next

        ctx.model = model
        ctx.dims = (B, T, R, S, d, H, dh, nL)
        ctx.drop = (p_drop, p_attn, seed_step)
        ctx.u_holds_grad = nL > 0 and u_holds_grad
        # (backward wants the REAL token ids where they exist - pad rows, the scatter into the vocabulary matrix; with
        #  text_embed there are none and it gets the row numbers)
        ctx.saved = (tok if text_embed is not None else x, totlen, rowmask, ximg16, loc, emb_saved, saved_layers)
        ctx.refine = (ref_saved, keylen_img, p_refine)
        ctx.input_grads = (R > 0 and x_img.requires_grad and track, text_embed is not None)
        ctx.langs = langs
        ctx.positions = positions
        ctx.h0_mode = None if h0 is None else h0.dtype
        # data parallelism: count the encoder passes that will be differentiated (only the last backward of a
        # step launches gradient buckets) and learn the token-row count the ranks pad to
        hook = model.ddp_hook
        ctx.track = bool(track) and hook is not None
        ctx.tok_rows_max = hook.encoder_forward(0 if h0 is not None else T * B) if ctx.track else None
        ctx.set_materialize_grads(False)
        # gradient sink of this pass's output (see GradSink): the heads add their rows' gradients into one zeroed [M, d]
        # buffer and hand autograd nothing; backward picks the buffer up here
        ctx.sink = GradSink()
        model._pending_sink = ctx.sink
        return h

    @staticmethod
    def backward(ctx, dout):
        model = ctx.model
        ar = model.arena()
        B, T, R, S, d, H, dh, nL = ctx.dims
        M = B * S
        p_drop, p_attn, seed_step = ctx.drop
        x, totlen, rowmask, ximg16, loc, emb_saved, saved_layers = ctx.saved
        ctx.saved = None
        seed = lambda kind, i=0: rng.stream_seed(model.base_seed, seed_step, _site(kind, i))   # noqa: E731
        ref_saved, keylen_img, p_refine = ctx.refine
        ctx.refine = None
        hook = model.ddp_hook if ctx.track else None
        if model.ddp_hook is not None:
            model.ddp_hook.params_ready(None)        # (the transposed weight copies the data gradients read)
        sunk = ctx.sink.take()
        if dout is None and sunk is None:
            if hook is not None:
                hook.encoder_backward_end()
            return (None,) * N_ENC_ARGS
        if dout is None:
            dh_ = sunk
        else:
            dh_ = dout.contiguous()
            if dh_.dtype != BF16:
                dh_ = dh_.to(BF16)
            if sunk is not None:          # a consumer outside the sink protocol handed autograd a gradient as well
                dh_ = dh_ + sunk.view_as(dh_)
        last = hook.encoder_backward_begin() if hook is not None else True
        st8 = model.fp8_state() if model.fp8 else None

        def dgrad(g, i, gsite, wsite, wt16, epi, pre8=None, **kw):
            """data gradient g [M, n] x W -> [M, k] on the transposed weight copy (bf8 gradient x fp8 weight when fp8 is on;
            pre8 = the gradient's 8-bit copy and its descale where the producing epilogue already left them)"""
            if st8 is None or wsite not in fp8mod.BWD_SITES:
                return ops.gemm_nt(g, wt16, epi, **kw)
            g8, dgs = pre8 if pre8 is not None else st8.quant(g, i, gsite)
            _, wt8, dws = st8.weights[(i, wsite)]
            return ops.gemm_nt_fp8(g8, wt8, epi, a_is_bf8=True, descale_a=dgs, descale_b=dws, **kw)

        for i in reversed(range(nL)):
            a, f = 'attentions.%d.' % i, 'ffns.%d.' % i
            (h_in, qkv, ctxt, lse, pre1, mean1, rstd1, x1, u, hact, pre2, mean2, rstd2) = saved_layers[i]
            saved_layers[i] = None
            # LayerNorm2 (+ the layer-end mask) and the FFN dropout
            dpre2, dY2 = ops.layernorm_bwd(dh_, None, pre2, ar.p('layer_norm2.%d.weight' % i), mean2, rstd2, rowmask,
                                           ar.g('layer_norm2.%d.weight' % i), ar.g('layer_norm2.%d.bias' % i),
                                           dbias_drop=ar.g(f + 'lin2.bias'), want_drop=p_drop > 0,
                                           seed=seed('ffn', i), p_drop=p_drop)
            if dY2 is None:
                dY2 = dpre2
            ops.gemm_wgrad(dY2, hact, ar.g(f + 'lin2.weight'))
            du8 = None
            if ctx.u_holds_grad == 'q':
                o8 = {}
                if st8 is not None and 'w1' in fp8mod.BWD_SITES:      # + the e5m2 copy of dU for the 8-bit dx1 product
                    kd = st8.index(i, 'du')
                    if st8.seen[kd]:
                        o8 = dict(out8=torch.empty((M, 4 * d), dtype=torch.uint8, device=dY2.device), scale8=st8.scale[kd:kd + 1],
                                  amax8=st8.amax[kd:kd + 1], out8_bf8=True)
                dU = ops.gemm_nt(dY2, ar.wt[('lin2', i)], L.EPI_MULQ, aux=u, colsum=ar.g(f + 'lin1.bias'), **o8)
                if o8:
                    du8 = (o8['out8'], st8.descale[kd:kd + 1])
                elif st8 is not None and 'w1' in fp8mod.BWD_SITES:
                    st8._first_use(dU, kd)
            else:
                dU = dgrad(dY2, i, 'dy2', 'w2', ar.wt[('lin2', i)], L.EPI_MUL if ctx.u_holds_grad else L.EPI_DGELU, aux=u,
                           colsum=ar.g(f + 'lin1.bias'))
            del hact, u, pre2
            ops.gemm_wgrad(dU, x1, ar.g(f + 'lin1.weight'))
            dx1 = dgrad(dU, i, 'du', 'w1', ar.wt[('lin1', i)], L.EPI_RES, aux=dpre2, pre8=du8)
            del dU, dpre2, dY2
            # LayerNorm1 and the attention-output dropout
            dpre1, dAO = ops.layernorm_bwd(dx1, None, pre1, ar.p('layer_norm1.%d.weight' % i), mean1, rstd1, None,
                                           ar.g('layer_norm1.%d.weight' % i), ar.g('layer_norm1.%d.bias' % i),
                                           dbias_drop=ar.g(a + 'out_lin.bias'), want_drop=p_drop > 0,
                                           seed=seed('attn_out', i), p_drop=p_drop)
            if dAO is None:
                dAO = dpre1
            dctx = dgrad(dAO, i, 'dao', 'wout', ar.wt[('out', i)], L.EPI_NONE)
            dqkv = ops.attn_bwd(qkv, totlen, ctxt, dctx, lse[0], B, S, H, dh, dbias_qkv=ar.qkv_bias(i, grad=True),
                                seed=seed('attn_p', i), p_drop=p_attn, keepmask=lse[1])
            # out_lin's and q/k/v's weight gradients in one launch (9 + 27 output tiles fill the CUs like one FFN gradient)
            ops.gemm_wgrad_pair(dqkv, h_in, ar.qkv_wgrad(i), dAO, ctxt, ar.g(a + 'out_lin.weight'))
            dh_ = dgrad(dqkv, i, 'dqkv', 'wqkv', ar.wt[('qkv', i)], L.EPI_RES, aux=dpre1)
            del dqkv, dctx, dAO, dpre1, dx1
            ar.touch_layer(i)
            if hook is not None:
                hook.layer_done(i, last)
        if ctx.h0_mode is not None:       # no embedding assembly in this pass: the rows' gradient goes back to whoever made them
            if hook is not None:
                hook.embed_done(last, ids=None, rows=None, n_max=ctx.tok_rows_max)
            return (None,) * 14 + (dh_.to(ctx.h0_mode), None)        # (h0 is the 15th argument, positions the 16th)
        # under data parallelism the token rows' gradients are exchanged as rows, not scattered here
        want_dximg, has_text_embed = ctx.input_grads
        langs, positions = ctx.langs, ctx.positions
        tok_rows = None
        if has_text_embed or langs is not None or positions is not None or (hook is not None and hook.active):
            tok_rows = torch.empty((T * B, d), dtype=BF16, device=dh_.device)
        grads = dict(
            d_g_emb=ar.g('layer_norm_emb.weight'), d_be_emb=ar.g('layer_norm_emb.bias'),
            d_pos=ar.g('position_embeddings.weight'), d_emb=ar.g('embeddings.weight'),
            d_g_img=ar.g('image_embeddings.LayerNorm.weight'), d_be_img=ar.g('image_embeddings.LayerNorm.bias'),
            d_b_img=ar.g('image_embeddings.image_embeddings.bias'),
            d_b_loc=ar.g('image_embeddings.image_location_embeddings.bias'),
            d_w_loc=ar.g('image_embeddings.image_location_embeddings.weight'))
        de = ops.embed_assemble_bwd(dh_, emb_saved, ar.p('layer_norm_emb.weight'),
                                    ar.p('image_embeddings.LayerNorm.weight'), x, totlen, loc, grads, B, T, R, d,
                                    -1 if has_text_embed else model.pad_index, seed_img=seed('img'), seed_emb=seed('emb'), p_drop=p_drop,
                                    img_rows_bwd=None if ref_saved is None else
                                    (lambda g: refiner_bwd(model, g, ref_saved, keylen_img, B, R, seed_step, p_refine)),
                                    tok_rows=tok_rows)
        ar.touch('layer_norm_emb.weight', 'layer_norm_emb.bias', 'position_embeddings.weight')
        if not has_text_embed:
            ar.touch('embeddings.weight')
        d_ximg = d_text = None
        if R > 0:
            ops.gemm_wgrad(de, ximg16, ar.g('image_embeddings.image_embeddings.weight'))
            ar.touch(*[n for n in ar.names if n.startswith('image_embeddings.')])
            if want_dximg:      # gradient wrt the region features (FreeLB's image_delta): de [R*B, d] @ W_img [d, 2048]
                d_ximg = ops.gemm_nt(de, _transposed(ar.w('image_embeddings.image_embeddings.weight')), L.EPI_NONE)
                d_ximg = d_ximg.view(R, B, 2048).float()
        if has_text_embed:
            d_text = tok_rows.view(T, B, d).transpose(0, 1).float()
            tok_rows = None
        if langs is not None:
            # the rows' gradients go to both tables: Lang[l] += sum of the rows with language l - a one-hot [T*B, n_langs]
            # matrix against the rows on the weight-gradient GEMM - and Emb[x] += row (below, or by the data-parallel
            # row exchange)
            npad = (model.n_langs + 7) // 8 * 8
            onehot = torch.zeros((T * B, npad), dtype=BF16, device=tok_rows.device)
            onehot.scatter_(1, langs.view(-1, 1), 1.0)
            dl = torch.zeros((npad, d), dtype=torch.float32, device=tok_rows.device)
            ops.gemm_wgrad(onehot, tok_rows, dl)
            ar.g('cross_lang_embeddings.weight').add_(dl[:model.n_langs])
            ar.touch('cross_lang_embeddings.weight')
        if positions is not None:       # the rows held P[pos] - P[t] (rows t * B + b): the kernel's own dP[t] is taken back
            gpos = ar.g('position_embeddings.weight')
            rows32 = tok_rows.float()
            gpos.index_add_(0, positions.reshape(-1), rows32)
            gpos[:T].sub_(rows32.view(T, B, d).sum(dim=1))
        if (langs is not None or positions is not None) and (hook is None or not hook.active):
            ops.scatter_add_token_rows(tok_rows, x.contiguous().view(-1), ar.g('embeddings.weight'), model.pad_index)
            tok_rows = None
        if hook is not None:
            hook.embed_done(last, ids=x if tok_rows is not None else None, rows=tok_rows, n_max=ctx.tok_rows_max)
        return (None, None, None, None, d_ximg, None, None, None, None, None, None, None, d_text, None, None, None)


class ImageStreamFn(torch.autograd.Function):
    """Input rows of crossfwd(stream_='img') (transformer.py:1044-1052, the encoder pass of the captioning step):
    BertImageEmbeddings (region projection + location projection -> LayerNorm -> its dropout, :247-269), + the language
    embedding, the stream's dropout, the length mask - no positions and no layer_norm_emb on this stream.
    x_img (R, B, 2048), image_loc (R, B, 5) -> bf16 [B*R, d] (rows b*R + r) for EncoderFn's layers-only mode."""

    @staticmethod
    def forward(ctx, anchor, model, x_img, lengths, image_loc, langs, p_drop, seed_step, p_refine=None, track=False):
        ar = model.arena()
        ar.refresh()
        dev = ar.device
        d = model.dim
        R, B = x_img.shape[0], x_img.shape[1]
        seed = lambda kind: rng.stream_seed(model.base_seed, seed_step, _site(kind))   # noqa: E731
        ximg16 = ops.cast_bf16(x_img.to(dev).contiguous().view(R * B, 2048))
        loc = image_loc.to(dev).contiguous().float()
        img_proj = ops.gemm_nt(ximg16, ar.w('image_embeddings.image_embeddings.weight'), L.EPI_BIAS,
                               bias=ar.p('image_embeddings.image_embeddings.bias'))
        rows, img_saved = ops.embed_image_rows_fwd(
            img_proj, loc, ar.p('image_embeddings.image_location_embeddings.weight'),
            ar.p('image_embeddings.image_location_embeddings.bias'), ar.p('image_embeddings.LayerNorm.weight'),
            ar.p('image_embeddings.LayerNorm.bias'), B, R, d, seed_img=seed('img'), p_drop=p_drop)
        if langs is not None:
            langs = langs.to(dev).contiguous()                                   # (R, B)
            rows = (rows.float() + ar.p('cross_lang_embeddings.weight')[langs.t()].reshape(B * R, d)).to(BF16)
        totlen = lengths.to(device=dev, dtype=torch.int32).contiguous()
        mask = (torch.arange(R, device=dev, dtype=torch.int32)[None, :] < totlen[:, None]).reshape(B * R, 1)
        h0 = ops.dropout_rows(rows, p_drop, seed('emb')) * mask.to(BF16)
        ref_saved = None
        if p_refine is not None:      # refine_image on this stream (transformer.py:1064-1066): the AoA refiner on the masked rows
            h0, ref_saved = refiner_fwd(model, h0.contiguous(), totlen, B, R, seed_step, p_refine)
        ctx.model = model
        ctx.saved = (ximg16, loc, img_saved, totlen, mask, langs, ref_saved)
        ctx.meta = (B, R, d, p_drop, seed_step, p_refine)
        ctx.x_img_meta = (x_img.dtype, x_img.device) if x_img.requires_grad else None      # (the FreeLB steps perturb the features)
        # data parallelism: this pass's gradients land in the 'embed' bucket after the encoder pass it feeds has run its
        # backward - the reducer must not launch that bucket before stream_backward_end()
        ctx.hook = model.ddp_hook if track else None
        if ctx.hook is not None:
            ctx.hook.stream_forward()
        return h0

    @staticmethod
    def backward(ctx, dh0):
        model = ctx.model
        ar = model.arena()
        ximg16, loc, img_saved, totlen, mask, langs, ref_saved = ctx.saved
        ctx.saved = None
        B, R, d, p_drop, seed_step, p_refine = ctx.meta
        seed = lambda kind: rng.stream_seed(model.base_seed, seed_step, _site(kind))   # noqa: E731
        if ref_saved is not None:
            dh0 = refiner_bwd(model, dh0.to(BF16).contiguous(), ref_saved, totlen, B, R, seed_step, p_refine)
            ar.touch(*[n for n in ar.names if n.startswith('refine_embeddings.')])
        g = (dh0.to(BF16) * mask.to(BF16)).contiguous()
        d_rows = ops.dropout_rows(g, p_drop, seed('emb'))
        if langs is not None:
            npad = (model.n_langs + 7) // 8 * 8
            onehot = torch.zeros((B * R, npad), dtype=BF16, device=g.device)
            onehot.scatter_(1, langs.t().reshape(-1, 1), 1.0)
            dl = torch.zeros((npad, d), dtype=torch.float32, device=g.device)
            ops.gemm_wgrad(onehot, d_rows, dl)
            ar.g('cross_lang_embeddings.weight').add_(dl[:model.n_langs])
            ar.touch('cross_lang_embeddings.weight')
        grads = dict(d_g_img=ar.g('image_embeddings.LayerNorm.weight'), d_be_img=ar.g('image_embeddings.LayerNorm.bias'),
                     d_b_img=ar.g('image_embeddings.image_embeddings.bias'),
                     d_b_loc=ar.g('image_embeddings.image_location_embeddings.bias'),
                     d_w_loc=ar.g('image_embeddings.image_location_embeddings.weight'))
        de = ops.embed_image_rows_bwd(d_rows, img_saved, ar.p('image_embeddings.LayerNorm.weight'), loc, totlen, grads, B, R, d,
                                      seed_img=seed('img'), p_drop=p_drop)
        ops.gemm_wgrad(de, ximg16, ar.g('image_embeddings.image_embeddings.weight'))
        ar.touch(*[n for n in ar.names if n.startswith('image_embeddings.')])
        d_ximg = None
        if ctx.x_img_meta is not None:
            # de rows are in the (r, b) order of ximg16, like in EncoderFn
            d_rows = ops.gemm_nt(de, _transposed(ar.w('image_embeddings.image_embeddings.weight')), L.EPI_NONE)
            d_ximg = d_rows.view(R, B, 2048).to(device=ctx.x_img_meta[1], dtype=ctx.x_img_meta[0])
        if ctx.hook is not None:
            ctx.hook.stream_backward_end()
        return (None, None, d_ximg) + (None,) * 7


N_DEC_ARGS = 12     # positional arguments of DecoderFn.forward


class DecoderFn(torch.autograd.Function):
    """crossfwd(stream_='text', causal=True, src_enc=..., src_len=...) with gradients: the teacher-forced pass of the
    translation / auto-encoding steps (transformer.py:1005-1102; caller xtrainer.py:1383-1441).  Text embedding assembly as
    in the non-causal stream, then per layer causal self-attention -> LN1 -> attention over the source encoding -> LN1.5
    -> FFN -> LN2.  Target sequences are short, so both attentions run on the rows kernels (csrc/decode.hip: a wave per
    (sequence, head, query)); every projection, LayerNorm and the embedding assembly are the encoder's kernels.  The
    gradient wrt src_enc is returned to autograd (it flows on into the encoder pass that produced it); parameter
    gradients go to the arena."""

    @staticmethod
    def forward(ctx, anchor, model, x, lengths, src_enc, src_len, langs, p_drop, p_attn, seed_step, positions=None, text_embed=None):
        ar = model.arena()
        ar.refresh()
        dev = ar.device
        d, H, nL = model.dim, model.n_heads, model.n_layers
        dh = d // H
        T, B = x.shape
        M = B * T
        dseed = lambda k, i=0: rng.stream_seed(model.base_seed, seed_step, _DEC_SITE0 + 8 * i + k)   # noqa: E731
        x = x.to(dev).contiguous()
        table, tok = ar.w('embeddings.weight'), x
        if langs is not None or positions is not None or text_embed is not None:
            # rows assembled here instead of gathered by the kernel: + the language embedding; explicit positions (the MASS
            # step decodes a span at its ORIGINAL positions, xtrainer.py:1684) as P[pos] - P[t], the kernel adds P[t] back
            # (text_embed: caller-made word rows (B, T, d) instead of Emb[x] - transformer.py:1053-1056, the FreeLB captioning step)
            rows = table[x.t()].float() if text_embed is None else text_embed.detach().to(dev).float()
            if langs is not None:
                langs = langs.to(dev).contiguous()
                rows = rows + ar.p('cross_lang_embeddings.weight')[langs.t()]
            if positions is not None:
                positions = positions.to(dev).contiguous()
                assert positions.size() == (T, B)
                ptab = ar.p('position_embeddings.weight')
                rows = rows + ptab[positions.t()] - ptab[:T][None, :, :]
            table = rows.to(BF16).reshape(B * T, d).contiguous()
            tok = ((torch.arange(B, device=dev, dtype=torch.int64) * T)[None, :] +
                   torch.arange(T, device=dev, dtype=torch.int64)[:, None]).contiguous()
        totlen = lengths.to(device=dev, dtype=torch.int32).contiguous()
        rowmask = (torch.arange(T, device=dev, dtype=torch.int32)[None, :] < totlen[:, None]).to(torch.uint8).contiguous().view(-1)
        h, emb_saved = ops.embed_assemble_fwd(
            tok, table, ar.p('position_embeddings.weight'), None, None,
            ar.p('image_embeddings.image_location_embeddings.weight'), ar.p('image_embeddings.image_location_embeddings.bias'),
            ar.p('image_embeddings.LayerNorm.weight'), ar.p('image_embeddings.LayerNorm.bias'),
            ar.p('layer_norm_emb.weight'), ar.p('layer_norm_emb.bias'), totlen, B, T, 0, d,
            seed_img=0, seed_emb=dseed(7), p_drop=p_drop)
        has_src = src_enc is not None
        S = src16 = src_klen = None
        if has_src:
            assert model.cross_attention_hot, 'the encoder-attention sub-layer is not in the training arena (params.mt_steps)'
            S = src_enc.shape[1]
            src16 = src_enc.detach().to(device=dev, dtype=BF16).contiguous().view(B * S, d)
            src_klen = src_len.to(device=dev, dtype=torch.int32).clamp(max=S).contiguous()
        qscale = 1.0 / math.sqrt(dh)
        saved = []
        for i in range(nL):
            a, f, e = 'attentions.%d.' % i, 'ffns.%d.' % i, 'encoder_attn.%d.' % i
            qkv = ops.gemm_nt(h, ar.qkv_w16(i), L.EPI_BIAS, bias=ar.qkv_bias(i), scale_cols=d, scale=qscale)
            ctxt, lse = ops.attn_rows_fwd(qkv, qkv.view(B, T, 3 * d)[:, :, d:], None, B, T, H, dh, T, causal=True,
                                          seed=dseed(0, i), p_drop=p_attn)
            pre1 = ops.gemm_nt(ctxt, ar.w(a + 'out_lin.weight'), L.EPI_BIAS_DROP_RES, bias=ar.p(a + 'out_lin.bias'), aux=h,
                               seed=dseed(1, i), p_drop=p_drop)
            x1, mean1, rstd1 = ops.layernorm_fwd(pre1, ar.p('layer_norm1.%d.weight' % i), ar.p('layer_norm1.%d.bias' % i))
            cross = None
            xf = x1
            if has_src:
                q2 = ops.gemm_nt(x1, ar.xattn(i, 'q'), L.EPI_BIAS, bias=ar.xattn(i, 'bq'), scale_cols=d, scale=qscale)
                kvc = ops.gemm_nt(src16, ar.xattn(i, 'kv'), L.EPI_BIAS, bias=ar.xattn(i, 'bkv')).view(B, S, 2 * d)
                ctx2, lse2 = ops.attn_rows_fwd(q2, kvc, src_klen, B, T, H, dh, S, seed=dseed(2, i), p_drop=p_attn)
                pre15 = ops.gemm_nt(ctx2, ar.w(e + 'out_lin.weight'), L.EPI_BIAS_DROP_RES, bias=ar.p(e + 'out_lin.bias'),
                                    aux=x1, seed=dseed(3, i), p_drop=p_drop)
                xf, mean15, rstd15 = ops.layernorm_fwd(pre15, ar.p('layer_norm15.%d.weight' % i), ar.p('layer_norm15.%d.bias' % i))
                cross = (q2, kvc, ctx2, lse2, pre15, mean15, rstd15)
            u = torch.empty((M, 4 * d), dtype=BF16, device=dev)
            hact = ops.gemm_nt(xf, ar.w(f + 'lin1.weight'), L.EPI_BIAS_GELU, bias=ar.p(f + 'lin1.bias'), out2=u)
            pre2 = ops.gemm_nt(hact, ar.w(f + 'lin2.weight'), L.EPI_BIAS_DROP_RES, bias=ar.p(f + 'lin2.bias'), aux=xf,
                               seed=dseed(4, i), p_drop=p_drop)
            h_next, mean2, rstd2 = ops.layernorm_fwd(pre2, ar.p('layer_norm2.%d.weight' % i), ar.p('layer_norm2.%d.bias' % i),
                                                     rowmask)
            saved.append((h, qkv, ctxt, lse, pre1, mean1, rstd1, x1, cross, xf, u, hact, pre2, mean2, rstd2))
            h = h_next
        ctx.model = model
        ctx.dims = (B, T, S, d, H, dh, nL)
        ctx.drop = (p_drop, p_attn, seed_step)
        ctx.saved = (x, totlen, rowmask, emb_saved, saved, src16, src_klen, langs, positions)
        ctx.text_meta = (text_embed.dtype, text_embed.device, text_embed.requires_grad) if text_embed is not None else None
        ctx.src_meta = (src_enc.dtype, src_enc.requires_grad) if has_src else None
        hook = model.ddp_hook
        ctx.track = hook is not None
        ctx.tok_rows_max = hook.encoder_forward(T * B) if ctx.track else None
        ctx.set_materialize_grads(False)
        return h

    @staticmethod
    def backward(ctx, dout):
        model = ctx.model
        ar = model.arena()
        B, T, S, d, H, dh, nL = ctx.dims
        p_drop, p_attn, seed_step = ctx.drop
        x, totlen, rowmask, emb_saved, saved, src16, src_klen, langs, positions = ctx.saved
        ctx.saved = None
        dseed = lambda k, i=0: rng.stream_seed(model.base_seed, seed_step, _DEC_SITE0 + 8 * i + k)   # noqa: E731
        hook = model.ddp_hook if ctx.track else None
        if dout is None:
            if hook is not None:
                hook.encoder_backward_end()
            return (None,) * N_DEC_ARGS
        dh_ = dout.contiguous()
        if dh_.dtype != BF16:
            dh_ = dh_.to(BF16)
        last = hook.encoder_backward_begin() if hook is not None else True
        qscale = 1.0 / math.sqrt(dh)
        has_src = src16 is not None
        d_src = None
        for i in reversed(range(nL)):
            a, f, e = 'attentions.%d.' % i, 'ffns.%d.' % i, 'encoder_attn.%d.' % i
            (h_in, qkv, ctxt, lse, pre1, mean1, rstd1, x1, cross, xf, u, hact, pre2, mean2, rstd2) = saved[i]
            saved[i] = None
            dpre2, dY2 = ops.layernorm_bwd(dh_, None, pre2, ar.p('layer_norm2.%d.weight' % i), mean2, rstd2, rowmask,
                                           ar.g('layer_norm2.%d.weight' % i), ar.g('layer_norm2.%d.bias' % i),
                                           dbias_drop=ar.g(f + 'lin2.bias'), want_drop=p_drop > 0, seed=dseed(4, i), p_drop=p_drop)
            if dY2 is None:
                dY2 = dpre2
            ops.gemm_wgrad(dY2, hact, ar.g(f + 'lin2.weight'))
            dU = ops.gemm_nt(dY2, ar.wt[('lin2', i)], L.EPI_DGELU, aux=u, colsum=ar.g(f + 'lin1.bias'))
            ops.gemm_wgrad(dU, xf, ar.g(f + 'lin1.weight'))
            dxf = ops.gemm_nt(dU, ar.wt[('lin1', i)], L.EPI_RES, aux=dpre2)
            if has_src:
                q2, kvc, ctx2, lse2, pre15, mean15, rstd15 = cross
                dpre15, dAO2 = ops.layernorm_bwd(dxf, None, pre15, ar.p('layer_norm15.%d.weight' % i), mean15, rstd15, None,
                                                 ar.g('layer_norm15.%d.weight' % i), ar.g('layer_norm15.%d.bias' % i),
                                                 dbias_drop=ar.g(e + 'out_lin.bias'), want_drop=p_drop > 0, seed=dseed(3, i),
                                                 p_drop=p_drop)
                if dAO2 is None:
                    dAO2 = dpre15
                ops.gemm_wgrad(dAO2, ctx2, ar.g(e + 'out_lin.weight'))
                dctx2 = ops.gemm_nt(dAO2, ar.wt[('xout', i)], L.EPI_NONE)
                dq2, dkvc = ops.attn_rows_bwd(q2, kvc, src_klen, dctx2, lse2, B, T, H, dh, S, qscale, seed=dseed(2, i), p_drop=p_attn)
                dkvc = dkvc.to(BF16).view(B * S, 2 * d)
                ops.gemm_wgrad(dq2, x1, ar.xattn(i, 'q', grad=True))
                ops.colsum(dq2, d, ar.xattn(i, 'bq', grad=True))
                ops.gemm_wgrad(dkvc, src16, ar.xattn(i, 'kv', grad=True))
                ops.colsum(dkvc, 2 * d, ar.xattn(i, 'bkv', grad=True))
                d_src = ops.gemm_nt(dkvc, ar.wt[('xkv', i)], L.EPI_NONE if d_src is None else L.EPI_RES, aux=d_src)
                dx1 = ops.gemm_nt(dq2, ar.wt[('xq', i)], L.EPI_RES, aux=dpre15)
            else:
                dx1 = dxf
            dpre1, dAO = ops.layernorm_bwd(dx1, None, pre1, ar.p('layer_norm1.%d.weight' % i), mean1, rstd1, None,
                                           ar.g('layer_norm1.%d.weight' % i), ar.g('layer_norm1.%d.bias' % i),
                                           dbias_drop=ar.g(a + 'out_lin.bias'), want_drop=p_drop > 0, seed=dseed(1, i), p_drop=p_drop)
            if dAO is None:
                dAO = dpre1
            ops.gemm_wgrad(dAO, ctxt, ar.g(a + 'out_lin.weight'))
            dctx = ops.gemm_nt(dAO, ar.wt[('out', i)], L.EPI_NONE)
            dqkv = torch.empty_like(qkv)
            _, dkv = ops.attn_rows_bwd(qkv, qkv.view(B, T, 3 * d)[:, :, d:], None, dctx, lse, B, T, H, dh, T, qscale, causal=True,
                                       seed=dseed(0, i), p_drop=p_attn, dq_out=dqkv)
            dqkv.view(B, T, 3 * d)[:, :, d:] = dkv
            ops.colsum(dqkv, 3 * d, ar.qkv_bias(i, grad=True))
            ops.gemm_wgrad(dqkv, h_in, ar.qkv_wgrad(i))
            dh_ = ops.gemm_nt(dqkv, ar.wt[('qkv', i)], L.EPI_RES, aux=dpre1)
            ar.touch_layer(i, cross=has_src)
            if hook is not None:
                hook.layer_done(i, last)
        tok_rows = None
        own_rows = langs is not None or positions is not None or ctx.text_meta is not None
        g_text = None
        if own_rows or (hook is not None and hook.active):
            tok_rows = torch.empty((T * B, d), dtype=BF16, device=dh_.device)
        grads = dict(
            d_g_emb=ar.g('layer_norm_emb.weight'), d_be_emb=ar.g('layer_norm_emb.bias'),
            d_pos=ar.g('position_embeddings.weight'), d_emb=ar.g('embeddings.weight'),
            d_g_img=ar.g('image_embeddings.LayerNorm.weight'), d_be_img=ar.g('image_embeddings.LayerNorm.bias'),
            d_b_img=ar.g('image_embeddings.image_embeddings.bias'),
            d_b_loc=ar.g('image_embeddings.image_location_embeddings.bias'),
            d_w_loc=ar.g('image_embeddings.image_location_embeddings.weight'))
        ops.embed_assemble_bwd(dh_, emb_saved, ar.p('layer_norm_emb.weight'), ar.p('image_embeddings.LayerNorm.weight'), x,
                               totlen, None, grads, B, T, 0, d, model.pad_index, seed_img=0, seed_emb=dseed(7), p_drop=p_drop,
                               tok_rows=tok_rows)
        ar.touch('layer_norm_emb.weight', 'layer_norm_emb.bias', 'position_embeddings.weight', 'embeddings.weight')
        if own_rows:
            if langs is not None:
                npad = (model.n_langs + 7) // 8 * 8
                onehot = torch.zeros((T * B, npad), dtype=BF16, device=tok_rows.device)
                onehot.scatter_(1, langs.view(-1, 1), 1.0)
                dl = torch.zeros((npad, d), dtype=torch.float32, device=tok_rows.device)
                ops.gemm_wgrad(onehot, tok_rows, dl)
                ar.g('cross_lang_embeddings.weight').add_(dl[:model.n_langs])
                ar.touch('cross_lang_embeddings.weight')
            if positions is not None:       # rows held P[pos] - P[t] (rows t * B + b): the kernel's own dP[t] is taken back
                gpos = ar.g('position_embeddings.weight')
                rows32 = tok_rows.float()
                gpos.index_add_(0, positions.reshape(-1), rows32)
                gpos[:T].sub_(rows32.view(T, B, d).sum(dim=1))
            if ctx.text_meta is not None:
                # the word rows were the caller's: their gradient goes back to autograd, nothing to the embedding matrix here
                assert hook is None or not hook.active, 'text_embed on the decoder is single-GPU (no token-row exchange for it)'
                if ctx.text_meta[2]:
                    g_text = tok_rows.view(T, B, d).transpose(0, 1).to(device=ctx.text_meta[1], dtype=ctx.text_meta[0])
                tok_rows = None
            elif hook is None or not hook.active:
                ops.scatter_add_token_rows(tok_rows, x.contiguous().view(-1), ar.g('embeddings.weight'), model.pad_index)
                tok_rows = None
        if hook is not None:
            hook.embed_done(last, ids=x if tok_rows is not None else None, rows=tok_rows, n_max=ctx.tok_rows_max)
        g_src = None
        if has_src and ctx.src_meta[1]:
            g_src = d_src.view(B, S, d).to(ctx.src_meta[0])
        return (None, None, None, None, g_src, None, None, None, None, None, None, g_text)


class MLMHeadFn(torch.autograd.Function):
    """predict() default branch: boolean-mask gather (transformer.py:1208), tied vocabulary
    projection (:111, :728-729) and mean cross-entropy (:112).  The bf16 logits are turned
    into their own gradient in place by the CE kernel, so backward is two GEMMs."""

    @staticmethod
    def forward(ctx, tensor, model, base, row_idx, y, scores_out, sink=None):
        ar = model.arena()
        ar.refresh()
        d, V = model.dim, model.n_words
        n = int(y.shape[0])
        ctx.sink = sink
        hsel = ops.gather_rows(base, row_idx, n, d)
        logits = torch.empty((n, ar.V_pad), dtype=BF16, device=hsel.device)
        full_tiles = _VOCAB_FULL_TILES and n >= 1024 and n % 256 == 0
        n_cols = ar.V_pad if full_tiles else V   # whole tiles: the eight-wave kernel
        fused_lse = _CE_FUSED_LSE and full_tiles
        stats = None
        if fused_lse:
            # the projection's epilogue also leaves (max, sum exp) of every row per 64-column block: the cross-entropy's
            # log-sum-exp becomes a reduction over V / 64 pairs instead of a pass over the 2.4 GB of logits
            # (the bias vector is read up to V_pad like the matrix: the arena bytes behind it)
            o = ar.offsets['pred_layer.proj.bias'][0]
            stats = torch.empty((ar.V_pad // 64, n, 2), dtype=torch.float32, device=hsel.device)
            ops.gemm_nt(hsel, ar.w('embeddings.weight'), L.EPI_BIAS_LSE, bias=ar.master[o:o + ar.V_pad], out=logits, n=n_cols,
                        out2=stats, scale_cols=V)
        else:
            ops.gemm_nt(hsel, ar.w('embeddings.weight'), L.EPI_BIAS, bias=ar.p('pred_layer.proj.bias'), out=logits, n=n_cols)
        if scores_out is not None:
            scores_out.append(logits[:, :V].float())
        dbias = None
        if fused_lse:
            loss_sum, _, dbias = ops.ce_from_block_stats(logits, V, y, stats, 1.0 / n, 1.0 / n)
        elif _VOCAB_FULL_TILES and n >= 1024:
            # the pass that writes the gradient also sums its columns (the output-bias gradient up to the upstream scale):
            # no second pass over 2.4 GB in backward
            loss_sum, _, dbias = ops.ce_fwd_bwd_colsum(logits, V, y, 1.0 / n, 1.0 / n)
        else:
            loss_sum, _ = ops.ce_fwd_bwd(logits, V, y, 1.0 / n, 1.0 / n)
        ctx.model = model
        ctx.saved = (hsel, logits, row_idx, tuple(tensor.shape), tuple(tensor.stride()), tensor.storage_offset(), base, dbias)
        return loss_sum[0].clone()

    @staticmethod
    def backward(ctx, gloss):
        model = ctx.model
        ar = model.arena()
        # first product of the step into the tied matrix's gradient (nothing has touched it, or the bias gradient its pad rows
        # alias, since the optimizer zeroed them): the weight gradient may store instead of adding
        # The store covers grad[o : o + V_pad * d]: the matrix AND (V_pad - V) * d elements behind it, so every parameter whose
        # gradient overlaps that range must be untouched too (with V ~ 95k and d = 1024 the pad rows run past the bias
        # gradient into position_embeddings' - which an accumulation micro-step without an MLM head may already have written).
        fresh = ar.range_untouched(ar.offsets['embeddings.weight'][0], ar.V_pad * model.dim)
        ar.touch('embeddings.weight', 'pred_layer.proj.bias')
        ar.vocab_stored = False
        d, V = model.dim, model.n_words
        hsel, dlogits, row_idx, shape, stride, soff, base, dbias = ctx.saved
        ctx.saved = None
        n = hsel.shape[0]
        g = gloss.reshape(1).float()
        hs = ops.scale_bf16_dev(hsel, g)
        if _VOCAB_FULL_TILES and n >= 4096 and n % 64 == 0:
            # whole 256-row tiles of the vocabulary: the four-wave weight-gradient kernel.  The pad columns of dlogits are
            # exact zeros (the CE kernel wrote them), so rows V .. V_pad - 1 of "the matrix" - the head of the bias
            # gradient that follows it in the arena - receive += 0
            o = ar.offsets['embeddings.weight'][0]
            will_store = fresh and L.load().m3p_gemm_wgrad_plan(n, ar.V_pad, d) == L.KERN_WGRAD_W4_TILES
            lazy = will_store and ar.stale == ar.vocab_range()      # (lazily zeroed by the optimizer: a STORE covers exactly that range)
            if not lazy:
                ar.ensure_zero()
            ar.vocab_stored = ops.gemm_wgrad(dlogits, hs, ar.grad[o:o + ar.V_pad * d].view(ar.V_pad, d), n=ar.V_pad, k=d,
                                             dw_is_zero=fresh, report_store=True, must_store=lazy)
            if lazy:
                # (only now: had the launcher declined the store, ops.gemm_wgrad has raised instead of accumulating onto the
                #  previous step's gradient - ADVICE r5; `stale` stays set until a store has really covered the range)
                ar.stale = None
            if will_store and not ar.vocab_stored:
                raise RuntimeError('the dispatch table promised a stored vocabulary weight gradient, the launcher accumulated')
        else:
            ops.gemm_wgrad(dlogits, hs, ar.g('embeddings.weight'), n=V, k=d)
        if dbias is not None:
            ops.axpy_dev(ar.g('pred_layer.proj.bias'), dbias[:V], g)
        else:
            ops.colsum(dlogits, V, ar.g('pred_layer.proj.bias'), scale=g)
        dH32 = torch.zeros((n, d), dtype=torch.float32, device=dlogits.device)
        # E [V, d] read in place (no transposed copy); whole 256-row tiles of predictions run on the four-wave kernel, which
        # reads all V_pad rows of "the matrix" - the bf16 arena behind E, finite numbers against dlogits' exact-zero pad columns
        ops.gemm_nn(dlogits, ar.w('embeddings.weight'), dH32, k_rows_readable=ar.V_pad if _VOCAB_DGRAD_W4 else None)
        dH = ops.scale_bf16_dev(dH32, g)
        # gradient wrt `tensor` (a strided view of the encoder output): the rows go to the pass's gradient sink, or - for a
        # tensor that is not an encoder pass's output - onto a zeroed twin of the underlying row buffer
        dtensor = _scatter_rows_grad(dH, row_idx, base, shape, stride, soff, ctx.sink)
        if model.ddp_hook is not None:
            model.ddp_hook.mlm_head_done()     # the dense part of the tied matrix is final: reduce it now
        return dtensor, None, None, None, None, None, None


def _as_row_buffer(tensor, d):
    """``tensor`` (.., d) - normally a strided view of the encoder output - as rows of its underlying
    contiguous [rows, d] buffer: returns (tensor, base, element strides of the leading dims, storage offset).
    Falls back to a contiguous copy when the view is not row-aligned."""
    strides, soff = tensor.stride(), tensor.storage_offset()
    with torch.no_grad():
        ok = strides[-1] == 1 and all(st % d == 0 for st in strides[:-1]) and soff % d == 0
        base = torch.as_strided(tensor, (tensor.untyped_storage().nbytes() // 2 // d, d), (d, 1), 0) if ok else None
    if base is None:
        tensor = tensor.contiguous()
        with torch.no_grad():
            base = tensor.view(-1, d)
        strides, soff = tensor.stride(), 0
    return tensor, base, strides, soff


def _scatter_rows_grad(dH, row_idx, base, shape, stride, soff, sink=None):
    """Gradient wrt a strided view of a row buffer: zeroed twin of the buffer + scatter-add of the selected rows - or,
    with the pass's GradSink, the rows go there and autograd gets nothing (GradSink)."""
    if sink is not None:
        ops.scatter_add_rows(dH, row_idx, sink.rows(base), dH.shape[0], dH.shape[1])
        return None
    dbase = torch.zeros_like(base)
    ops.scatter_add_rows(dH, row_idx, dbase, dH.shape[0], dH.shape[1])
    return torch.as_strided(dbase, shape, stride, soff)


class ObjHeadFn(torch.autograd.Function):
    """Masked-region classification (MRM): predict(is_obj=True), transformer.py:1205-1210 =
    BertPredictionHeadTransform (:595-606: dense, erf-GELU, LayerNorm eps 1e-12) + ObjPredLayer (:575-584:
    Linear(d, 1600), mean CE with ignore_index = -1).  Only the masked rows (label != -1) are gathered and
    pushed through the head - the others do not enter the ignore_index mean.  GEMMs + LayerNorm + CE
    are the kernels of the encoder / MLM head; GELU backward is m3p_gelu_bwd."""

    @staticmethod
    def forward(ctx, tensor, model, base, row_idx, y, sink=None):
        ar = model.arena()
        ar.refresh()
        d = model.dim
        n = int(y.shape[0])
        ctx.sink = sink
        hsel = ops.gather_rows(base, row_idx, n, d)
        u = torch.empty((n, d), dtype=BF16, device=hsel.device)
        t = ops.gemm_nt(hsel, ar.w('transformer_obj.dense.weight'), L.EPI_BIAS_GELU, bias=ar.p('transformer_obj.dense.bias'), out2=u)
        x1, mean, rstd = ops.layernorm_fwd(t, ar.p('transformer_obj.LayerNorm.weight'), ar.p('transformer_obj.LayerNorm.bias'))
        logits = ops.gemm_nt(x1, ar.w('pred_obj_layer.proj.weight'), L.EPI_BIAS, bias=ar.p('pred_obj_layer.proj.bias'))
        loss_sum, _ = ops.ce_fwd_bwd(logits, logits.shape[1], y, 1.0 / n, 1.0 / n)
        ctx.model = model
        ctx.saved = (hsel, u, t, mean, rstd, x1, logits, row_idx, tuple(tensor.shape), tuple(tensor.stride()),
                     tensor.storage_offset(), base)
        return loss_sum[0].clone()

    @staticmethod
    def backward(ctx, gloss):
        model = ctx.model
        ar = model.arena()
        hsel, u, t, mean, rstd, x1, dlogits, row_idx, shape, stride, soff, base = ctx.saved
        ctx.saved = None
        n, d = hsel.shape
        g = gloss.reshape(1).float()
        dev = hsel.device
        ar.touch('transformer_obj.dense.weight', 'transformer_obj.dense.bias', 'transformer_obj.LayerNorm.weight',
                 'transformer_obj.LayerNorm.bias', 'pred_obj_layer.proj.weight', 'pred_obj_layer.proj.bias')
        nobj = dlogits.shape[1]
        ops.gemm_wgrad(dlogits, (x1.float() * g).to(BF16), ar.g('pred_obj_layer.proj.weight'))
        ops.colsum(dlogits, nobj, ar.g('pred_obj_layer.proj.bias'), scale=g)
        dx1 = torch.zeros((n, d), dtype=torch.float32, device=dev)
        ops.gemm_nn_streamk(dlogits, ar.w('pred_obj_layer.proj.weight'), dx1)
        dx1 = (dx1 * g).to(BF16)
        dt, _ = ops.layernorm_bwd(dx1, None, t, ar.p('transformer_obj.LayerNorm.weight'), mean, rstd, None,
                                  ar.g('transformer_obj.LayerNorm.weight'), ar.g('transformer_obj.LayerNorm.bias'))
        du = ops.gelu_bwd(dt, u)
        ops.gemm_wgrad(du, hsel, ar.g('transformer_obj.dense.weight'))
        ops.colsum(du, d, ar.g('transformer_obj.dense.bias'))
        dH = torch.zeros((n, d), dtype=torch.float32, device=dev)
        ops.gemm_nn_streamk(du, ar.w('transformer_obj.dense.weight'), dH)
        return _scatter_rows_grad(dH.to(BF16), row_idx, base, shape, stride, soff, ctx.sink), None, None, None, None, None


class MrfrHeadFn(torch.autograd.Function):
    """Masked-region feature regression (MRFR): predict(is_mrfr=True) = mrfr_dense (transformer.py:1202-1204)
    on the masked regions + F.mse_loss against their original 2048-d features (xtrainer.py:2332-2352)."""

    @staticmethod
    def forward(ctx, tensor, model, base, row_idx, target, sink=None):
        ar = model.arena()
        ar.refresh()
        d = model.dim
        n = int(target.shape[0])
        ctx.sink = sink
        hsel = ops.gather_rows(base, row_idx, n, d)
        reg = ops.gemm_nt(hsel, ar.w('mrfr_dense.weight'), L.EPI_BIAS, bias=ar.p('mrfr_dense.bias'))
        sq, dreg = ops.mse_fwd_bwd(reg, target, 1.0 / (n * reg.shape[1]))
        ctx.model = model
        ctx.saved = (hsel, dreg, row_idx, tuple(tensor.shape), tuple(tensor.stride()), tensor.storage_offset(), base)
        return (sq[0] / (n * reg.shape[1])).clone()

    @staticmethod
    def backward(ctx, gloss):
        model = ctx.model
        ar = model.arena()
        hsel, dreg, row_idx, shape, stride, soff, base = ctx.saved
        ctx.saved = None
        n, d = hsel.shape
        g = gloss.reshape(1).float()
        ar.touch('mrfr_dense.weight', 'mrfr_dense.bias')
        ops.gemm_wgrad(dreg, ops.scale_bf16_dev(hsel, g), ar.g('mrfr_dense.weight'))
        ops.colsum(dreg, dreg.shape[1], ar.g('mrfr_dense.bias'), scale=g)
        dH = torch.zeros((n, d), dtype=torch.float32, device=hsel.device)
        ops.gemm_nn_streamk(dreg, ar.w('mrfr_dense.weight'), dH)
        return _scatter_rows_grad(ops.scale_bf16_dev(dH, g), row_idx, base, shape, stride, soff, ctx.sink), None, None, None, None, None


class DenseRowsFn(torch.autograd.Function):
    """y = x W^T + b for one of the arena's Linear layers on arbitrary bf16 rows [n, d_in] (predict(is_mrfr=True),
    transformer.py:1202-1204).  GEMM forward, column-sum / weight-gradient / data-gradient GEMMs backward."""

    @staticmethod
    def forward(ctx, x, model, wname, bname):
        ar = model.arena()
        ar.refresh()
        ctx.model, ctx.names = model, (wname, bname)
        ctx.save_for_backward(x)
        return ops.gemm_nt(x, ar.w(wname), L.EPI_BIAS, bias=ar.p(bname))

    @staticmethod
    def backward(ctx, dy):
        model = ctx.model
        ar = model.arena()
        wname, bname = ctx.names
        x, = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != BF16:
            dy = dy.to(BF16)
        ar.touch(wname, bname)
        ops.colsum(dy, dy.shape[1], ar.g(bname))
        ops.gemm_wgrad(dy, x, ar.g(wname))
        dx = ops.gemm_nt(dy, _transposed(ar.w(wname)), L.EPI_NONE)
        return dx, None, None, None


def mrfr_dense_rows(model, tensor):
    """predict(is_mrfr=True): mrfr_dense on every row of tensor (..., d) -> (..., 2048) bf16."""
    d = model.dim
    assert tensor.shape[-1] == d
    x = tensor.to(BF16).contiguous().view(-1, d)
    return DenseRowsFn.apply(x, model, 'mrfr_dense.weight', 'mrfr_dense.bias').view(*tensor.shape[:-1], -1)


def _masked_region_rows(tensor, labels, d):
    """labels (B*R,) with -1 = not masked (host tensor preferred: no device sync) -> (tensor, base, int32 row
    indices of the masked (b, r) positions inside the row buffer, their labels on the device)."""
    assert tensor.dim() == 3 and tensor.shape[-1] == d and tensor.dtype == BF16
    B, R, _ = tensor.shape
    tensor, base, strides, soff = _as_row_buffer(tensor, d)
    lab = labels.reshape(-1)
    pos = torch.nonzero(lab.cpu() != -1).view(-1)        # host side: labels come from the data pipeline
    b_idx, r_idx = pos // R, pos % R
    row_idx = ((soff + b_idx * strides[0] + r_idx * strides[1]) // d).to(torch.int32).to(tensor.device)
    return tensor, base, row_idx, pos


def mrm_head(model, tensor, y_all):
    """predict(is_obj=True): tensor (B, R, d) bf16 image part of the encoder output, y_all (B*R,) int64."""
    tensor, base, row_idx, pos = _masked_region_rows(tensor, y_all, model.dim)
    assert pos.numel() > 0, 'no masked region in the batch'
    y = y_all.reshape(-1).cpu()[pos].to(tensor.device)
    return ObjHeadFn.apply(tensor, model, base, row_idx, y, _sink_of(tensor, base))


def mrfr_head(model, tensor, obj_labels, ori_att_feats):
    """MRFR loss of xtrainer.py:2332-2352 on the masked regions of tensor (B, R, d)."""
    tensor, base, row_idx, pos = _masked_region_rows(tensor, obj_labels, model.dim)
    if pos.numel() == 0:
        return torch.zeros((), dtype=torch.float32, device=tensor.device)
    feats = ori_att_feats.reshape(-1, ori_att_feats.shape[-1])
    target = feats[pos.to(feats.device)].to(device=tensor.device, dtype=torch.float32).contiguous()
    return MrfrHeadFn.apply(tensor, model, base, row_idx, target, _sink_of(tensor, base))


class ItmHeadFn(torch.autograd.Function):
    """BertPooler + relation head: (pooled_layer, seq_relationship) for ITM (transformer.py:546-558, :1194-1197)
    or (pooled_layer2, seq_relationship2) for the CLCM pass (:1198-1201).  The d x d products run on the bf16
    GEMMs, tanh / score / derivative glue in csrc/itm.hip.  Parameter gradients accumulate straight into the
    gradient arena (main-grad)."""

    @staticmethod
    def forward(ctx, first, model, pooler='pooled_layer', rel='seq_relationship', sink=None, base=None, row_idx=None):
        ar = model.arena()
        ar.refresh()
        ctx.sink = (sink, base, row_idx)
        h16, pooled, scores = ops.itm_head_fwd(first, ar.w(pooler + '.dense.weight'), ar.p(pooler + '.dense.bias').detach(),
                                               ar.p(rel + '.weight').detach().view(-1), ar.p(rel + '.bias').detach())
        ctx.model = model
        ctx.names = (pooler, rel)
        ctx.saved = (h16, pooled)
        ctx.set_materialize_grads(False)
        return scores.view(-1, 1)

    @staticmethod
    def backward(ctx, dscores):
        if dscores is None:
            return (None,) * 7
        model = ctx.model
        ar = model.arena()
        pooler, rel = ctx.names
        h16, pooled = ctx.saved
        ctx.saved = None
        ds = dscores.reshape(-1).float().contiguous()
        dh = ops.itm_head_bwd(ds, h16, pooled, ar.w(pooler + '.dense.weight'), ar.p(rel + '.weight').detach().view(-1),
                              ar.g(pooler + '.dense.weight'), ar.g(pooler + '.dense.bias'),
                              ar.g(rel + '.weight').view(-1), ar.g(rel + '.bias'))
        ar.touch(pooler + '.dense.weight', pooler + '.dense.bias', rel + '.weight', rel + '.bias')
        sink, base, row_idx = ctx.sink
        if sink is not None:       # the B first-position rows go to the pass's gradient sink (GradSink)
            ops.scatter_add_rows(dh, row_idx, sink.rows(base), dh.shape[0], dh.shape[1])
            dh = None
        return (dh,) + (None,) * 6


class ItmLossFn(torch.autograd.Function):
    """xtrainer.py:2357-2372 on the device in one launch: multi_cls_loss_weight * CE over groups of sample_n scores +
    bin_cls_loss_weight * BCE-with-logits against the one-hot labels (the reference moves the scores to the CPU for
    this).  The kernel returns the loss and its gradient together."""

    @staticmethod
    def forward(ctx, scores, pos, sample_n, w_ce, w_bce):
        sc = scores.detach().reshape(-1)
        if sc.dtype != torch.float32:
            sc = sc.float()
        loss, dsc = ops.itm_loss_fwd_bwd(sc.contiguous(), pos, int(sample_n), w_ce, w_bce)
        ctx.dsc, ctx.meta = dsc, (tuple(scores.shape), scores.dtype)
        return loss.view(())

    @staticmethod
    def backward(ctx, gloss):
        shape, dtype = ctx.meta
        d = ctx.dsc * gloss.reshape(1).float()
        ctx.dsc = None
        return d.view(shape).to(dtype), None, None, None, None


def mlm_head(model, tensor, pred_mask, y, want_scores):
    """Resolve ``tensor`` (T,B,d) — normally the view ``encoder_out[R:]`` — to rows of its
    underlying contiguous [rows, d] buffer, so the gather reads the activation in place.
    (The reference's host-side ``assert (y == pad).sum().item() == 0`` (:108) is a device
    sync per step and is not reproduced.)"""
    d = model.dim
    assert tensor.dim() == 3 and tensor.shape[-1] == d and tensor.dtype == BF16, \
        'predict() expects the bf16 encoder output (T, B, d)'
    T, B, _ = tensor.shape
    s0, s1, s2 = tensor.stride()
    soff = tensor.storage_offset()
    with torch.no_grad():
        if s2 == 1 and s0 % d == 0 and s1 % d == 0 and soff % d == 0:
            base = torch.as_strided(tensor, (tensor.untyped_storage().nbytes() // 2 // d, d), (d, 1), 0)
        else:
            base = None
    if base is None:
        tensor = tensor.contiguous()
        with torch.no_grad():
            base = tensor.view(T * B, d)
        s0, s1, soff = B * d, d, 0
    n = int(y.shape[0])
    pm = pred_mask.to(tensor.device)
    if pm.dtype not in (torch.bool, torch.uint8):
        pm = pm != 0
    # rows of the True entries in (t, b) order: one compaction launch, no host sync (the count is y's length)
    row_idx = ops.mask_to_rows(pm.contiguous(), B, s0, s1, soff, d, n)
    scores_out = [] if want_scores else None
    loss = MLMHeadFn.apply(tensor, model, base, row_idx, y.to(tensor.device), scores_out, _sink_of(tensor, base))
    return loss, (scores_out[0] if want_scores else None)
