"""Decoder inference: ``crossfwd(causal=True, src_enc=..., cache=...)`` and the two search loops built on
it, ``generate`` (greedy / sampled) and ``generate_beam`` - transformer.py:970-1114 (the causal branch with the
``encoder_attn`` / ``layer_norm15`` sub-layer of :1087-1091), :149-210 (attention with the key / value cache),
:1216-1317, :1319-1515, :1518-1561.

One decoding step is latency / HBM work: one new token per sequence, every weight read once.  The projections run on the
bf16 GEMM of the training path (fused bias, 1/sqrt(dh), residual and GELU epilogues), attention on
``m3p_attn_query_fwd`` (csrc/decode.hip: one wave per (sequence, head, query) over the cached keys / values).  The cache
holds, per layer, ONE token-major bf16 tensor [bs, capacity, 2 d] (keys | values) for the self-attention and one
[bs, S_src, 2 d] for the encoder attention (projected once, at the first step); the reference keeps (k, v) head-major
tuples under the module ids and concatenates per step.  ``cache['slen']`` has the reference's meaning.

This module is forward only; the teacher-forced training pass of the same stream is ``functional.DecoderFn``
(``TransformerModel.crossfwd`` picks it in training mode), and calling ``decoder_forward`` itself with autograd enabled on
a model in training mode raises.
"""
import heapq
import math

import torch

from . import lib as L
from . import ops

BF16 = torch.bfloat16


class _ColdWeights:
    """bf16 working copies of the encoder-attention sub-layer (parameters outside the training arena), re-made when any of
    them changed."""

    def __init__(self, model):
        self.model = model
        self.key = None
        self.q, self.kv, self.out, self.bq, self.bkv, self.bo = [], [], [], [], [], []

    def refresh(self):
        m = self.model
        ps = []
        for i in range(m.n_layers):
            for lin in ('q_lin', 'k_lin', 'v_lin', 'out_lin'):
                mod = m.get_submodule('encoder_attn.%d.%s' % (i, lin))
                ps += [mod.weight, mod.bias]
        key = tuple((p._version, p.data_ptr()) for p in ps)
        if key == self.key:
            return self
        self.q, self.kv, self.out, self.bq, self.bkv, self.bo = [], [], [], [], [], []
        for i in range(m.n_layers):
            g = lambda lin: m.get_submodule('encoder_attn.%d.%s' % (i, lin))   # noqa: E731
            self.q.append(g('q_lin').weight.detach().to(BF16).contiguous())
            self.kv.append(torch.cat([g('k_lin').weight.detach(), g('v_lin').weight.detach()]).to(BF16).contiguous())
            self.out.append(g('out_lin').weight.detach().to(BF16).contiguous())
            self.bq.append(g('q_lin').bias.detach().float().contiguous())
            self.bkv.append(torch.cat([g('k_lin').bias.detach(), g('v_lin').bias.detach()]).float().contiguous())
            self.bo.append(g('out_lin').bias.detach().float().contiguous())
        self.key = key
        return self


def _self_cache(cache, i, bs, need, d, dev):
    """The layer's self-attention key/value tensor with room for `need` positions (grown by doubling)."""
    key = ('self', i)
    cur = cache.get(key)
    if cur is None or cur.shape[1] < need:
        cap = max(int(cache.get('max_len', 0)), 64, need)
        if cur is not None:
            cap = max(cap, 2 * cur.shape[1])
        new = torch.zeros((bs, cap, 2 * d), dtype=BF16, device=dev)
        if cur is not None:
            new[:, :cur.shape[1]] = cur
        cache[key] = cur = new
    return cur


def decoder_forward(model, x, lengths, src_enc=None, src_len=None, positions=None, langs=None, cache=None):
    """crossfwd(stream_='text', causal=True): x (slen, bs) int64 -> (n_new, bs, d) bf16, n_new = slen - cache['slen'] (all of
    them without a cache).  src_enc (bs, S, d) / src_len (bs) switch the encoder-attention sub-layer on."""
    if torch.is_grad_enabled() and model.training:
        raise NotImplementedError('the causal decoder is built for inference (eval mode / torch.no_grad()): its '
                                  'teacher-forced training step is not part of this build (SURVEY 8 f4)')
    assert (src_enc is None) == (src_len is None)
    slen, bs = x.size()
    assert lengths.size(0) == bs
    d, H = model.dim, model.n_heads
    dh = d // H
    dev = model.embeddings.weight.device
    ar = model.arena()
    ar.refresh()
    pos0 = int(cache['slen']) if cache is not None else 0
    n_new = slen - pos0
    assert n_new >= 1
    x = x.to(dev)
    lengths = lengths.to(dev)
    if positions is None:
        positions = torch.arange(slen, device=dev)[:, None].expand(slen, bs)
    else:
        assert positions.size() == (slen, bs)
        positions = positions.to(dev)
    tok = x[pos0:].t()                                                     # (bs, n_new), the reference's x[:, -_slen:]
    h = ar.w('embeddings.weight')[tok].float() + model.position_embeddings.weight.detach()[positions[pos0:].t()]
    if langs is not None:
        assert langs.size() == (slen, bs)
        h = h + model.cross_lang_embeddings.weight.detach()[langs.to(dev)[pos0:].t()]
    rowmask = (torch.arange(pos0, slen, device=dev)[None, :] < lengths[:, None]).to(torch.uint8).reshape(-1).contiguous()
    h16, _, _ = ops.layernorm_fwd(h.to(BF16).reshape(bs * n_new, d).contiguous(), model.layer_norm_emb.weight.detach(),
                                  model.layer_norm_emb.bias.detach(), rowmask=rowmask)
    qscale = 1.0 / math.sqrt(dh)
    cw = None
    if src_enc is not None:
        assert src_enc.size(0) == bs and src_enc.size(2) == d
        S = src_enc.size(1)
        cw = model.decoder_cold_weights()
        src_klen = src_len.to(dev).to(torch.int32).clamp(max=S).contiguous()
        src16 = None
    for i in range(model.n_layers):
        a, f = 'attentions.%d.' % i, 'ffns.%d.' % i
        wqkv, bqkv = ar.qkv_w16(i), ar.qkv_bias(i)
        q = ops.gemm_nt(h16, wqkv[:d], L.EPI_BIAS, bias=bqkv[:d], scale_cols=d, scale=qscale)
        kv = ops.gemm_nt(h16, wqkv[d:], L.EPI_BIAS, bias=bqkv[d:]).view(bs, n_new, 2 * d)
        if cache is not None:
            store = _self_cache(cache, i, bs, slen, d, dev)
            store[:, pos0:slen] = kv
            kv = store
        ctx = ops.attn_query_fwd(q, kv, None, bs, n_new, H, dh, slen, causal=True, pos0=pos0)
        pre = ops.gemm_nt(ctx, ar.w(a + 'out_lin.weight'), L.EPI_BIAS_DROP_RES, bias=ar.p(a + 'out_lin.bias'), aux=h16)
        h16, _, _ = ops.layernorm_fwd(pre, ar.p('layer_norm1.%d.weight' % i), ar.p('layer_norm1.%d.bias' % i))
        if cw is not None:
            q2 = ops.gemm_nt(h16, cw.q[i], L.EPI_BIAS, bias=cw.bq[i], scale_cols=d, scale=qscale)
            kvc = cache.get(('cross', i)) if cache is not None else None
            if kvc is None:
                if src16 is None:
                    src16 = src_enc.detach().to(device=dev, dtype=BF16).contiguous().view(bs * S, d)
                kvc = ops.gemm_nt(src16, cw.kv[i], L.EPI_BIAS, bias=cw.bkv[i]).view(bs, S, 2 * d)
                if cache is not None:
                    cache[('cross', i)] = kvc
            ctx2 = ops.attn_query_fwd(q2, kvc, src_klen, bs, n_new, H, dh, S)
            pre = ops.gemm_nt(ctx2, cw.out[i], L.EPI_BIAS_DROP_RES, bias=cw.bo[i], aux=h16)
            ln15 = model.get_submodule('layer_norm15.%d' % i)
            h16, _, _ = ops.layernorm_fwd(pre, ln15.weight.detach(), ln15.bias.detach())
        u = torch.empty((bs * n_new, 4 * d), dtype=BF16, device=dev)
        hact = ops.gemm_nt(h16, ar.w(f + 'lin1.weight'), L.EPI_BIAS_GELU, bias=ar.p(f + 'lin1.bias'), out2=u)
        pre = ops.gemm_nt(hact, ar.w(f + 'lin2.weight'), L.EPI_BIAS_DROP_RES, bias=ar.p(f + 'lin2.bias'), aux=h16)
        h16, _, _ = ops.layernorm_fwd(pre, ar.p('layer_norm2.%d.weight' % i), ar.p('layer_norm2.%d.bias' % i), rowmask=rowmask)
    if cache is not None:
        cache['slen'] = pos0 + n_new
    return h16.view(bs, n_new, d).transpose(0, 1)


def word_scores(model, tensor):
    """PredLayer.get_scores (transformer.py:120-124): (n, d) -> (n, n_words) fp32 on the tied vocabulary matrix."""
    ar = model.arena()
    ar.refresh()
    V = model.n_words
    x16 = tensor.detach().to(BF16).reshape(-1, model.dim).contiguous()
    logits = torch.empty((x16.shape[0], ar.V_pad), dtype=BF16, device=x16.device)
    ops.gemm_nt(x16, ar.w('embeddings.weight'), L.EPI_BIAS, bias=ar.p('pred_layer.proj.bias'), out=logits, n=V)
    return logits[:, :V].float()


def generate(model, src_enc, src_len, tgt_lang_id, max_len=200, sample_temperature=None):
    """transformer.py:1216-1317: greedy (or temperature-sampled) decoding with the key / value cache.
    -> (generated (cur_len, bs) int64, gen_len (bs))."""
    bs = len(src_len)
    assert src_enc.size(0) == bs
    dev = model.embeddings.weight.device
    src_len = src_len.to(dev)
    generated = torch.full((max_len, bs), model.pad_index, dtype=torch.long, device=dev)
    generated[0].fill_(model.eos_index)                       # <EOS> doubles as <BOS>
    positions = torch.arange(max_len, device=dev)[:, None].expand(max_len, bs)
    langs = None
    if tgt_lang_id is not None:
        langs = torch.full((max_len, bs), int(tgt_lang_id), dtype=torch.long, device=dev)
    cur_len = 1
    gen_len = torch.ones(bs, dtype=torch.long, device=dev)
    unfinished = torch.ones(bs, dtype=torch.long, device=dev)
    cache = {'slen': 0, 'max_len': max_len}
    while cur_len < max_len:
        tensor = decoder_forward(model, generated[:cur_len], gen_len, src_enc, src_len, positions[:cur_len],
                                 None if langs is None else langs[:cur_len], cache)
        assert tensor.size() == (1, bs, model.dim)
        scores = word_scores(model, tensor[-1])
        if sample_temperature is None:
            next_words = torch.topk(scores, 1)[1].squeeze(1)
        else:
            next_words = torch.multinomial(torch.softmax(scores / sample_temperature, dim=1), 1).squeeze(1)
        generated[cur_len] = next_words * unfinished + model.pad_index * (1 - unfinished)
        gen_len.add_(unfinished)
        unfinished.mul_(next_words.ne(model.eos_index).long())
        cur_len += 1
        if int(unfinished.max()) == 0:          # (one host read per step, as in the reference)
            break
    if cur_len == max_len:
        generated[-1].masked_fill_(unfinished.bool(), model.eos_index)
    assert int((generated == model.eos_index).sum()) == 2 * bs
    return generated[:cur_len], gen_len


class BeamHypotheses(object):
    """The n best finished hypotheses of one sentence, ranked by length-normalised log-probability (the bookkeeping of
    transformer.py:1518-1561).  A bounded min-heap keyed on (score, arrival order): the root is the entry the next better
    hypothesis evicts, `worst_score` is the root's score; among equal scores the earliest arrival goes first, which is
    the tie-break of the reference's sort."""

    def __init__(self, n_hyp, max_len, length_penalty, early_stopping):
        self.n_hyp, self.early_stopping = n_hyp, early_stopping
        self.length_penalty = length_penalty
        self._norm = float(max_len - 1) ** length_penalty      # longest possible hypothesis (without <BOS>)
        self._heap, self._arrivals = [], 0

    def __len__(self):
        return len(self._heap)

    @property
    def hyp(self):
        """[(score, tokens)] in arrival order."""
        return [(s, t) for s, _, t in sorted(self._heap, key=lambda e: e[1])]

    @property
    def worst_score(self):
        return self._heap[0][0] if self._heap else 1e9

    def best(self):
        """Tokens of the best hypothesis (the earliest among equals)."""
        return max(self._heap, key=lambda e: (e[0], -e[1]))[2]

    def add(self, hyp, sum_logprobs):
        entry = (sum_logprobs / len(hyp) ** self.length_penalty, self._arrivals, hyp)
        self._arrivals += 1
        if len(self._heap) < self.n_hyp:
            heapq.heappush(self._heap, entry)
        elif entry[0] > self._heap[0][0]:
            heapq.heapreplace(self._heap, entry)

    def is_done(self, best_sum_logprobs):
        """Can no open beam still enter the list?  (always, once it is full, under early stopping)"""
        full = len(self._heap) >= self.n_hyp
        return full and (self.early_stopping or self.worst_score >= best_sum_logprobs / self._norm)


def generate_beam(model, src_enc, src_len, tgt_lang_id, beam_size, length_penalty, early_stopping, max_len=200):
    """transformer.py:1319-1515: beam search; the beam is folded into the batch dimension (bs * beam_size rows), the key /
    value caches are re-ordered by the surviving beams' source rows after every step.
    -> (decoded (max tgt_len, bs) int64, tgt_len (bs))."""
    assert src_enc.size(0) == src_len.size(0) and beam_size >= 1
    bs = len(src_len)
    n_words = model.n_words
    dev = model.embeddings.weight.device
    src_len = src_len.to(dev)
    src_enc = src_enc.to(dev).unsqueeze(1).expand((bs, beam_size) + src_enc.shape[1:]).contiguous().view(
        (bs * beam_size,) + src_enc.shape[1:])
    src_len = src_len.unsqueeze(1).expand(bs, beam_size).contiguous().view(-1)
    generated = torch.full((max_len, bs * beam_size), model.pad_index, dtype=torch.long, device=dev)
    generated[0].fill_(model.eos_index)
    hyps = [BeamHypotheses(beam_size, max_len, length_penalty, early_stopping) for _ in range(bs)]
    positions = torch.arange(max_len, device=dev)[:, None].expand_as(generated)
    # (the reference always builds language ids here, :1370 - a model without language embeddings cannot take them)
    langs = positions.clone().fill_(int(tgt_lang_id)) if tgt_lang_id is not None else None
    beam_scores = torch.zeros((bs, beam_size), dtype=torch.float32, device=dev)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(-1)
    cur_len = 1
    cache = {'slen': 0, 'max_len': max_len}
    done = [False] * bs
    while cur_len < max_len:
        lengths = torch.full((bs * beam_size,), cur_len, dtype=torch.long, device=dev)
        tensor = decoder_forward(model, generated[:cur_len], lengths, src_enc, src_len, positions[:cur_len],
                                 None if langs is None else langs[:cur_len], cache)
        assert tensor.size() == (1, bs * beam_size, model.dim)
        scores = torch.log_softmax(word_scores(model, tensor[-1]), dim=-1)
        _scores = (scores + beam_scores[:, None]).view(bs, beam_size * n_words)
        next_scores, next_words = torch.topk(_scores, 2 * beam_size, dim=1, largest=True, sorted=True)
        next_scores_h, next_words_h = next_scores.tolist(), next_words.tolist()       # one host copy per step
        next_batch_beam = []
        for sent in range(bs):
            done[sent] = done[sent] or hyps[sent].is_done(max(next_scores_h[sent]))
            if done[sent]:
                next_batch_beam.extend([(0, model.pad_index, 0)] * beam_size)
                continue
            next_sent_beam = []
            for idx, value in zip(next_words_h[sent], next_scores_h[sent]):
                beam_id, word_id = idx // n_words, idx % n_words
                if word_id == model.eos_index or cur_len + 1 == max_len:
                    hyps[sent].add(generated[:cur_len, sent * beam_size + beam_id].clone(), value)
                else:
                    next_sent_beam.append((value, word_id, sent * beam_size + beam_id))
                if len(next_sent_beam) == beam_size:
                    break
            assert len(next_sent_beam) == (0 if cur_len + 1 == max_len else beam_size)
            if len(next_sent_beam) == 0:
                next_sent_beam = [(0, model.pad_index, 0)] * beam_size
            next_batch_beam.extend(next_sent_beam)
        assert len(next_batch_beam) == bs * beam_size
        beam_scores = torch.tensor([v[0] for v in next_batch_beam], dtype=torch.float32, device=dev)
        beam_words = torch.tensor([v[1] for v in next_batch_beam], dtype=torch.long, device=dev)
        beam_idx = torch.tensor([v[2] for v in next_batch_beam], dtype=torch.long, device=dev)
        generated = generated[:, beam_idx]
        generated[cur_len] = beam_words
        for k in list(cache.keys()):
            if isinstance(k, tuple):                       # key / value tensors follow their beams
                cache[k] = cache[k].index_select(0, beam_idx)
        cur_len += 1
        if all(done):
            break
    tgt_len = torch.zeros(bs, dtype=torch.long, device=dev)
    best = []
    for i, hp in enumerate(hyps):
        best_hyp = hp.best()
        tgt_len[i] = len(best_hyp) + 1                      # + <EOS>
        best.append(best_hyp)
    decoded = torch.full((int(tgt_len.max()), bs), model.pad_index, dtype=torch.long, device=dev)
    for i, hypo in enumerate(best):
        decoded[:int(tgt_len[i]) - 1, i] = hypo
        decoded[int(tgt_len[i]) - 1, i] = model.eos_index
    assert int((decoded == model.eos_index).sum()) == 2 * bs
    return decoded, tgt_len
