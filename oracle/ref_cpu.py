"""ORACLE — test infrastructure only.  NOT part of the product path.

A plain PyTorch (CPU, fp32 or fp64) restatement of the algorithm on M3P's
pre-training hot path, written from the reference's behaviour; every function cites the
reference file:line it follows (paths relative to /root/reference).  It is the checker
for the HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  Nothing under ``m3p_amd/`` imports it.

Pinning: ``tests/test_oracle_golden.py`` checks every function here against golden
vectors produced by importing the reference itself in the development container
(``oracle/gen_goldens.py`` -> ``tests/golden/*.npz``).  The reference ships no tests or
known-answer vectors of its own, so those generated vectors are the pin.

Functional style: parameters come in as a ``dict`` keyed by the reference's state-dict
names, so the same dict can be loaded into the reference model, this oracle and the
HIP model.  Dropout is expressed through optional *keep masks* (already scaled or
not, see ``_drop``) so a test can feed the exact masks the HIP kernels generate.
"""
import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-12  # nn.LayerNorm(dim, eps=1e-12): M3P/src/model/transformer.py:244,660,694,709


# ----------------------------------------------------------------------------
# elementary pieces
# ----------------------------------------------------------------------------

def gelu_erf(x):
    """M3P/src/model/transformer.py:48-56 — exact (erf) GELU."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def layer_norm(x, w, b, eps=LN_EPS):
    """nn.LayerNorm over the last dim, biased variance (transformer.py:694,709)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def _drop(x, keep, p):
    """F.dropout with an explicit keep mask: y = x * keep / (1 - p).
    ``keep`` None or p == 0 -> identity (eval mode / parity runs)."""
    if keep is None or p == 0:
        return x
    return x * keep.to(x.dtype) / (1.0 - p)


def get_masks(slen, lengths):
    """transformer.py:59-78, non-causal branch: mask[b, s] = s < lengths[b];
    the attention mask is the same (bs, slen) tensor."""
    assert int(lengths.max()) <= slen
    alen = torch.arange(slen, dtype=torch.long)
    mask = alen[None, :] < lengths[:, None]
    return mask, mask


def multi_head_attention(x, mask, wq, bq, wk, bk, wv, bv, wo, bo, n_heads,
                         p_attn=0.0, keep_attn=None, return_ctx=False):
    """transformer.py:149-210, self-attention branch (kv=None, cache=None).

    x (bs, S, d); mask (bs, S) bool, True = valid key.  q is scaled by 1/sqrt(d_h)
    *after* the bias (:197); padded keys get -inf (:199-200); softmax in fp32 (:202);
    dropout on the probabilities (:203); context = P @ v (:204); out_lin (:208)."""
    bs, S, d = x.shape
    dh = d // n_heads

    def shape(t):
        return t.view(bs, S, n_heads, dh).transpose(1, 2)

    q = shape(F.linear(x, wq, bq)) / math.sqrt(dh)
    k = shape(F.linear(x, wk, bk))
    v = shape(F.linear(x, wv, bv))
    scores = torch.matmul(q, k.transpose(2, 3))
    scores = scores.masked_fill(~mask[:, None, None, :], float('-inf'))
    w = torch.softmax(scores.float(), dim=-1).to(scores.dtype)
    w = _drop(w, keep_attn, p_attn)
    ctx = torch.matmul(w, v).transpose(1, 2).contiguous().view(bs, S, d)
    out = F.linear(ctx, wo, bo)
    if return_ctx:
        return out, ctx
    return out


def transformer_ffn(x, w1, b1, w2, b2, p=0.0, keep=None):
    """transformer.py:222-227: lin2(gelu(lin1(x))) then dropout."""
    return _drop(F.linear(gelu_erf(F.linear(x, w1, b1)), w2, b2), keep, p)


def image_embeddings(sd, x_img, loc, p=0.0, keep=None, prefix='image_embeddings.'):
    """BertImageEmbeddings.forward, transformer.py:247-269 with input_dist=None (the
    only way jointfwd calls it, :901): LN(W_img x + b + W_loc loc + b) then dropout."""
    e = F.linear(x_img, sd[prefix + 'image_embeddings.weight'], sd[prefix + 'image_embeddings.bias'])
    e = e + F.linear(loc, sd[prefix + 'image_location_embeddings.weight'],
                     sd[prefix + 'image_location_embeddings.bias'])
    e = layer_norm(e, sd[prefix + 'LayerNorm.weight'], sd[prefix + 'LayerNorm.bias'])
    return _drop(e, keep, p)


# ----------------------------------------------------------------------------
# the encoder: TransformerModel.jointfwd
# ----------------------------------------------------------------------------

def aoa_refiner(sd, x, attn_mask, n_refine_layers, n_heads, p=0.0, keeps=None, prefix='refine_embeddings.'):
    """AoA_Refiner_Core.forward, transformer.py:410-422 (built at :662; applied to the image rows by
    jointfwd(refine_image=True), :905-906).  x (B, R, d), attn_mask (B, R) bool -> (B, R, d).
    Per layer (AoA_Refiner_Layer :405-407 over pre-norm SublayerConnection :392-394):
      x = x + drop(AoA(LN_a(x)))  with AoA = MultiHeadedDotAttention(do_aoa=1) :327-371:
            q, k, v = linears[0..2](xn)  -> heads of d_k = d / h                      (:352-354)
            P = dropout(softmax(q k^T / sqrt(d_k), masked_fill(mask == 0, -inf)))      (attention_sub :274-284)
            out = GLU(Linear_{2d -> 2d}(dropout_aoa(cat([P v, xn], -1))))              (:364-366)
      x = x + drop(FFN(LN_b(x)))  with TransformerFFN (:222-227: lin2(gelu(lin1)), then its own dropout)
    then the final LayerNorm (:422).  All dropout rates are the constructor defaults 0.1 in the reference
    (:288, :411 - not params.dropout); ``p`` here, with optional keep masks
    keeps[('ref_attn_p', i) | ('ref_aoa', i) | ('ref_sub0', i) | ('ref_ffn', i) | ('ref_sub1', i)]."""
    keeps = keeps or {}
    B, R, d = x.shape
    dk = d // n_heads

    def heads(t):
        return t.view(B, R, n_heads, dk).transpose(1, 2)

    for i in range(n_refine_layers):
        pre = prefix + 'layers.%d.' % i
        xn = layer_norm(x, sd[pre + 'sublayer.0.norm.weight'], sd[pre + 'sublayer.0.norm.bias'])
        q, k, v = [heads(F.linear(xn, sd[pre + 'self_attn.linears.%d.weight' % j], sd[pre + 'self_attn.linears.%d.bias' % j]))
                   for j in range(3)]
        scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dk)
        scores = scores.masked_fill(~attn_mask[:, None, None, :], -float('inf'))
        pa = _drop(F.softmax(scores, dim=-1), keeps.get(('ref_attn_p', i)), p)
        att = torch.matmul(pa, v).transpose(1, 2).contiguous().view(B, R, d)
        cat = _drop(torch.cat([att, xn], -1), keeps.get(('ref_aoa', i)), p)
        ab = F.linear(cat, sd[pre + 'self_attn.aoa_layer.0.weight'], sd[pre + 'self_attn.aoa_layer.0.bias'])
        x = x + _drop(F.glu(ab, dim=-1), keeps.get(('ref_sub0', i)), p)
        xn = layer_norm(x, sd[pre + 'sublayer.1.norm.weight'], sd[pre + 'sublayer.1.norm.bias'])
        f = transformer_ffn(xn, sd[pre + 'feed_forward.lin1.weight'], sd[pre + 'feed_forward.lin1.bias'],
                            sd[pre + 'feed_forward.lin2.weight'], sd[pre + 'feed_forward.lin2.bias'],
                            p=p, keep=keeps.get(('ref_ffn', i)))
        x = x + _drop(f, keeps.get(('ref_sub1', i)), p)
    return layer_norm(x, sd[prefix + 'norm.weight'], sd[prefix + 'norm.bias'])


def jointfwd(sd, n_layers, n_heads, x, lengths, x_img, lengths_img, image_loc,
             dropout=0.0, attention_dropout=0.0, keeps=None, text_embed=None, refine_layers=0, refine_dropout=0.0):
    """TransformerModel.jointfwd, transformer.py:878-968 (refine_layers > 0: with refine_image=True).

    x (T, B) int64; x_img (R, B, 2048); image_loc (R, B, 5) -> (S=R+T, B, d).
    Order of operations reproduced exactly:
      img = image_embeddings(...)                                     (:901)
      tok = Emb[x] (pad row is zero by construction)                  (:913)
      h = cat([img, tok], dim=1); h += Pos[0..S-1]                    (:929-936)
      h *= mask (valid prefix of length len_img+len_txt, :917-919)    (:940)
      h = LN_emb(h); dropout                                          (:942-943)
      per layer: h = LN1(h + drop(attn(h))); h = LN2(h + ffn(h)); h *= mask   (:947-958)
    ``keeps``: optional dict of keep masks {'img','emb', ('attn_p', i), ('attn_out', i), ('ffn', i)}.
    """
    keeps = keeps or {}
    T, B = x.shape
    xt = x.t()
    img = image_embeddings(sd, x_img.transpose(0, 1), image_loc.transpose(0, 1),
                           p=dropout, keep=keeps.get('img'))
    R = img.shape[1]
    if refine_layers:      # refine_image=True (:903-906): the refiner sees the image rows under their own length mask
        _, img_attn_mask = get_masks(R, lengths_img)
        img = aoa_refiner(sd, img, img_attn_mask, refine_layers, n_heads, p=refine_dropout, keeps=keeps)
    tok = text_embed if text_embed is not None else F.embedding(xt, sd['embeddings.weight'])
    S = R + T
    mask, attn_mask = get_masks(S, lengths_img + lengths)
    h = torch.cat([img, tok], dim=1)
    h = h + sd['position_embeddings.weight'][:S][None]
    h = h * mask[..., None].to(h.dtype)
    h = layer_norm(h, sd['layer_norm_emb.weight'], sd['layer_norm_emb.bias'])
    h = _drop(h, keeps.get('emb'), dropout)
    for i in range(n_layers):
        a = 'attentions.%d.' % i
        attn = multi_head_attention(
            h, attn_mask,
            sd[a + 'q_lin.weight'], sd[a + 'q_lin.bias'], sd[a + 'k_lin.weight'], sd[a + 'k_lin.bias'],
            sd[a + 'v_lin.weight'], sd[a + 'v_lin.bias'], sd[a + 'out_lin.weight'], sd[a + 'out_lin.bias'],
            n_heads, p_attn=attention_dropout, keep_attn=keeps.get(('attn_p', i)))
        attn = _drop(attn, keeps.get(('attn_out', i)), dropout)
        h = layer_norm(h + attn, sd['layer_norm1.%d.weight' % i], sd['layer_norm1.%d.bias' % i])
        f = 'ffns.%d.' % i
        h = h + transformer_ffn(h, sd[f + 'lin1.weight'], sd[f + 'lin1.bias'],
                                sd[f + 'lin2.weight'], sd[f + 'lin2.bias'],
                                p=dropout, keep=keeps.get(('ffn', i)))
        h = layer_norm(h, sd['layer_norm2.%d.weight' % i], sd['layer_norm2.%d.bias' % i])
        h = h * mask[..., None].to(h.dtype)
    return h.transpose(0, 1)


def crossfwd_text(sd, n_layers, n_heads, x, lengths, dropout=0.0, attention_dropout=0.0, keeps=None, langs=None,
                  positions=None):
    """TransformerModel.crossfwd(stream_='text', causal=False, positions=None, langs=None):
    transformer.py:1050-1102 — the text-only stream behind Trainer.mlm_step (xtrainer.py:757).
    Differs from jointfwd in the order at the input: Emb[x] + Pos -> LN_emb -> dropout -> *mask
    (:1055-1062), then the same post-LN layers.  x (T, B) -> (T, B, d)."""
    keeps = keeps or {}
    T, B = x.shape
    mask, attn_mask = get_masks(T, lengths)
    if positions is None:       # :1011-1014
        pos = sd['position_embeddings.weight'][:T][None]
    else:                       # explicit (T, B) positions: TLM batches restart them at the second sentence (:1057-1058)
        pos = F.embedding(positions.t(), sd['position_embeddings.weight'])
    h = F.embedding(x.t(), sd['embeddings.weight']) + pos
    if langs is not None:       # :1059-1060 (multilingual models: n_langs > 1)
        h = h + F.embedding(langs.t(), sd['cross_lang_embeddings.weight'])
    h = layer_norm(h, sd['layer_norm_emb.weight'], sd['layer_norm_emb.bias'])
    h = _drop(h, keeps.get('emb'), dropout)
    h = h * mask[..., None].to(h.dtype)
    for i in range(n_layers):
        a = 'attentions.%d.' % i
        attn = multi_head_attention(
            h, attn_mask,
            sd[a + 'q_lin.weight'], sd[a + 'q_lin.bias'], sd[a + 'k_lin.weight'], sd[a + 'k_lin.bias'],
            sd[a + 'v_lin.weight'], sd[a + 'v_lin.bias'], sd[a + 'out_lin.weight'], sd[a + 'out_lin.bias'],
            n_heads, p_attn=attention_dropout, keep_attn=keeps.get(('attn_p', i)))
        attn = _drop(attn, keeps.get(('attn_out', i)), dropout)
        h = layer_norm(h + attn, sd['layer_norm1.%d.weight' % i], sd['layer_norm1.%d.bias' % i])
        f = 'ffns.%d.' % i
        h = h + transformer_ffn(h, sd[f + 'lin1.weight'], sd[f + 'lin1.bias'], sd[f + 'lin2.weight'],
                                sd[f + 'lin2.bias'], p=dropout, keep=keeps.get(('ffn', i)))
        h = layer_norm(h, sd['layer_norm2.%d.weight' % i], sd['layer_norm2.%d.bias' % i])
        h = h * mask[..., None].to(h.dtype)
    return h.transpose(0, 1)


# ----------------------------------------------------------------------------
# causal decoder: crossfwd(causal=True, src_enc=...), greedy decoding
# ----------------------------------------------------------------------------
def attention_kv(x, kv, mask, sd, prefix, n_heads):
    """transformer.py:149-210 with a separate key / value source: x (bs, Tq, d) queries, kv (bs, Tk, d), mask (bs, Tk) or
    (bs, Tq, Tk) bool (True = attend).  Without a cache (recomputing every position each step gives what the cached
    run gives: a cached key / value is the projection of the same hidden state)."""
    bs, Tq, d = x.shape
    Tk = kv.shape[1]
    dh = d // n_heads
    q = F.linear(x, sd[prefix + 'q_lin.weight'], sd[prefix + 'q_lin.bias']).view(bs, Tq, n_heads, dh).transpose(1, 2)
    k = F.linear(kv, sd[prefix + 'k_lin.weight'], sd[prefix + 'k_lin.bias']).view(bs, Tk, n_heads, dh).transpose(1, 2)
    v = F.linear(kv, sd[prefix + 'v_lin.weight'], sd[prefix + 'v_lin.bias']).view(bs, Tk, n_heads, dh).transpose(1, 2)
    scores = torch.matmul(q / math.sqrt(dh), k.transpose(2, 3))
    m4 = mask[:, None, :, :] if mask.dim() == 3 else mask[:, None, None, :]
    scores = scores.masked_fill(~m4, float('-inf'))
    w = torch.softmax(scores.float(), dim=-1).to(scores.dtype)
    ctx = torch.matmul(w, v).transpose(1, 2).contiguous().view(bs, Tq, d)
    return F.linear(ctx, sd[prefix + 'out_lin.weight'], sd[prefix + 'out_lin.bias'])


def decoder_crossfwd(sd, n_layers, n_heads, x, lengths, src_enc=None, src_len=None, positions=None, langs=None, enc_mask=None,
                     text_embed=None):
    """TransformerModel.crossfwd(stream_='text', causal=True, src_enc, src_len) in eval mode, transformer.py:1005-1102:
    mask[b, s] = s < lengths[b]; the causal attention mask is position-only (:70-71: key <= query, padded keys inside
    the window ARE attended); Emb[x] + Pos (+ Lang) -> LN_emb -> * mask; per layer self-attention -> LN1 ->
    encoder attention over src_enc[:, :src_len] -> LN1.5 -> FFN -> LN2 -> * mask.  x (T, B) -> (T, B, d)."""
    T, B = x.shape
    alen = torch.arange(T)
    mask = alen[None, :] < lengths[:, None]
    attn_mask = (alen[None, None, :] <= alen[None, :, None]).expand(B, T, T)
    pos = alen[None, :].expand(B, T) if positions is None else positions.t()
    tok = F.embedding(x.t(), sd['embeddings.weight']) if text_embed is None else text_embed      # (:1053-1056)
    h = tok + F.embedding(pos, sd['position_embeddings.weight'])
    if langs is not None:
        h = h + F.embedding(langs.t(), sd['cross_lang_embeddings.weight'])
    h = layer_norm(h, sd['layer_norm_emb.weight'], sd['layer_norm_emb.bias'])
    h = h * mask[..., None].to(h.dtype)
    if src_enc is not None:
        src_mask = torch.arange(int(src_len.max()))[None, :] < src_len[:, None]
        if enc_mask is not None:          # :1016-1017: source positions the decoder must not look at (the MASS step's <mask>s)
            src_mask = src_mask & enc_mask
    for i in range(n_layers):
        h = layer_norm(h + attention_kv(h, h, attn_mask, sd, 'attentions.%d.' % i, n_heads),
                       sd['layer_norm1.%d.weight' % i], sd['layer_norm1.%d.bias' % i])
        if src_enc is not None:
            h = layer_norm(h + attention_kv(h, src_enc[:, :src_mask.shape[1]], src_mask, sd, 'encoder_attn.%d.' % i, n_heads),
                           sd['layer_norm15.%d.weight' % i], sd['layer_norm15.%d.bias' % i])
        f = 'ffns.%d.' % i
        h = h + transformer_ffn(h, sd[f + 'lin1.weight'], sd[f + 'lin1.bias'], sd[f + 'lin2.weight'], sd[f + 'lin2.bias'])
        h = layer_norm(h, sd['layer_norm2.%d.weight' % i], sd['layer_norm2.%d.bias' % i])
        h = h * mask[..., None].to(h.dtype)
    return h.transpose(0, 1)


def crossfwd_img(sd, n_layers, n_heads, x_img, lengths, image_loc, langs=None, n_refine_layers=0):
    """TransformerModel.crossfwd(stream_='img', causal=False) in eval mode, transformer.py:1044-1102: BertImageEmbeddings on the
    region features (+ the language embedding), * mask - no positions and no layer_norm_emb on this stream - then the
    post-LN layers.  x_img (R, B, 2048), image_loc (R, B, 5) -> (R, B, d)."""
    R, B = x_img.shape[0], x_img.shape[1]
    mask, attn_mask = get_masks(R, lengths)
    h = image_embeddings(sd, x_img.transpose(0, 1), image_loc.transpose(0, 1))
    if langs is not None:
        h = h + F.embedding(langs.t(), sd['cross_lang_embeddings.weight'])
    h = h * mask[..., None].to(h.dtype)
    if n_refine_layers:            # refine_image=True on this stream (:1064-1066)
        h = aoa_refiner(sd, h, attn_mask, n_refine_layers, n_heads)
    for i in range(n_layers):
        a = 'attentions.%d.' % i
        attn = multi_head_attention(
            h, attn_mask,
            sd[a + 'q_lin.weight'], sd[a + 'q_lin.bias'], sd[a + 'k_lin.weight'], sd[a + 'k_lin.bias'],
            sd[a + 'v_lin.weight'], sd[a + 'v_lin.bias'], sd[a + 'out_lin.weight'], sd[a + 'out_lin.bias'], n_heads)
        h = layer_norm(h + attn, sd['layer_norm1.%d.weight' % i], sd['layer_norm1.%d.bias' % i])
        f = 'ffns.%d.' % i
        h = h + transformer_ffn(h, sd[f + 'lin1.weight'], sd[f + 'lin1.bias'], sd[f + 'lin2.weight'], sd[f + 'lin2.bias'])
        h = layer_norm(h, sd['layer_norm2.%d.weight' % i], sd['layer_norm2.%d.bias' % i])
        h = h * mask[..., None].to(h.dtype)
    return h.transpose(0, 1)


def word_scores(sd, h):
    """PredLayer.get_scores with the tied matrix (transformer.py:120-124, :728-729)."""
    return F.linear(h, sd['embeddings.weight'], sd['pred_layer.proj.bias'])


def greedy_decode(sd, n_layers, n_heads, src_enc, src_len, tgt_lang_id, max_len, pad_index=1, eos_index=2):
    """TransformerModel.generate, greedy branch (transformer.py:1216-1317), every step recomputed from scratch over the
    prefix (no cache).  Also returns, per step, the margin between the two best scores of every unfinished sentence -
    tests use it to tell a real mismatch from a near-tie flipped by bf16 arithmetic."""
    bs = len(src_len)
    generated = torch.full((max_len, bs), pad_index, dtype=torch.long)
    generated[0] = eos_index
    gen_len = torch.ones(bs, dtype=torch.long)
    unfinished = torch.ones(bs, dtype=torch.long)
    cur_len, margins = 1, []
    while cur_len < max_len:
        langs = None if tgt_lang_id is None else torch.full((cur_len, bs), int(tgt_lang_id), dtype=torch.long)
        h = decoder_crossfwd(sd, n_layers, n_heads, generated[:cur_len], gen_len, src_enc, src_len, langs=langs)
        # (the cached run evaluates position cur_len - 1 when gen_len has just reached cur_len for unfinished sentences)
        scores = word_scores(sd, h[-1])
        top2 = scores.topk(2, dim=1)[0]
        margins.append(torch.where(unfinished.bool(), top2[:, 0] - top2[:, 1], torch.full((bs,), float('inf'))))
        nxt = scores.argmax(dim=1)
        generated[cur_len] = nxt * unfinished + pad_index * (1 - unfinished)
        gen_len += unfinished
        unfinished = unfinished * nxt.ne(eos_index).long()
        cur_len += 1
        if int(unfinished.max()) == 0:
            break
    if cur_len == max_len:
        generated[-1].masked_fill_(unfinished.bool(), eos_index)
    return generated[:cur_len], gen_len, torch.stack(margins)


# ----------------------------------------------------------------------------
# heads: TransformerModel.predict
# ----------------------------------------------------------------------------

def predict_mlm(sd, tensor, pred_mask, y, pad_index=1):
    """predict (default branch) + PredLayer.forward: transformer.py:1208-1212, 104-117.
    tensor (T, B, d) sequence-major text part; pred_mask (T, B) bool; boolean gather
    gives rows in (s, b) order; scores = rows @ Emb^T + b_V (tied weight, :728-729);
    loss = mean cross-entropy.  Returns (scores, loss)."""
    assert int((y == pad_index).sum()) == 0
    d = tensor.shape[-1]
    rows = tensor[pred_mask.unsqueeze(-1).expand_as(tensor)].view(-1, d)
    scores = F.linear(rows, sd['embeddings.weight'], sd['pred_layer.proj.bias'])
    loss = F.cross_entropy(scores.float(), y, reduction='mean')
    return scores, loss


def predict_relation(sd, tensor):
    """predict(is_relation=True): transformer.py:1194-1197 + BertPooler :546-558.
    tensor (B, S, d) batch-major; pools position 0 (image region 0) -> (B, 1)."""
    pooled = torch.tanh(F.linear(tensor[:, 0], sd['pooled_layer.dense.weight'], sd['pooled_layer.dense.bias']))
    return F.linear(pooled, sd['seq_relationship.weight'], sd['seq_relationship.bias'])


def bert_head_transform(sd, x, prefix='transformer_obj.'):
    """BertPredictionHeadTransform.forward, transformer.py:595-606: LayerNorm(gelu(dense(x))), eps 1e-12."""
    h = gelu_erf(F.linear(x, sd[prefix + 'dense.weight'], sd[prefix + 'dense.bias']))
    return layer_norm(h, sd[prefix + 'LayerNorm.weight'], sd[prefix + 'LayerNorm.bias'])


def predict_obj(sd, tensor, y):
    """predict(is_obj=True), transformer.py:1205-1210 + ObjPredLayer.forward :575-584 (masked region
    modelling): tensor (B, R, d) image part of the encoder output, batch-major; y (B*R,) object class of the
    masked regions, -1 elsewhere; scores over 1600 classes, mean CE with ignore_index = -1."""
    h = bert_head_transform(sd, tensor)
    scores = F.linear(h, sd['pred_obj_layer.proj.weight'], sd['pred_obj_layer.proj.bias']).view(-1, 1600)
    loss = F.cross_entropy(scores.float(), y, reduction='mean', ignore_index=-1)
    return scores, loss


def predict_mrfr(sd, tensor):
    """predict(is_mrfr=True), transformer.py:1202-1204: mrfr_dense(tensor) -> (B, R, 2048)."""
    return F.linear(tensor, sd['mrfr_dense.weight'], sd['mrfr_dense.bias'])


def mrfr_loss(reg, obj_labels, ori_att_feats):
    """xtrainer.py:2332-2352: MSE between the regressed and the original features of the masked regions."""
    mask = obj_labels.reshape(-1) != -1
    pred = reg.reshape(-1, 2048)[mask]
    tgt = ori_att_feats.reshape(-1, 2048)[mask]
    if pred.shape[0] == 0:
        return torch.zeros((), dtype=reg.dtype)
    return F.mse_loss(pred.float(), tgt.float())


def predict_clcm(sd, tensor):
    """predict(is_clcm=True): transformer.py:1198-1201 - the second BertPooler + Linear(d, 1) on the joint
    encoding of the images with the OTHER caption (xtrainer.py:2379-2393)."""
    pooled = torch.tanh(F.linear(tensor[:, 0], sd['pooled_layer2.dense.weight'], sd['pooled_layer2.dense.bias']))
    return F.linear(pooled, sd['seq_relationship2.weight'], sd['seq_relationship2.bias'])


def clcm_loss(sd, n_layers, n_heads, batch, x2, len2, clcm_labels):
    """The CLCM second pass of pretrain_under_step (i2t task, xtrainer.py:2379-2393): jointfwd of the same
    regions with caption x2, relation scores through the second head, BCE-with-logits against clcm_labels."""
    out2 = jointfwd(sd, n_layers, n_heads, x2, len2, batch['x_img'], batch['lengths_img'], batch['image_loc'])
    rel2 = predict_clcm(sd, out2.transpose(0, 1))
    return F.binary_cross_entropy_with_logits(rel2.view(-1).float(), clcm_labels.view(-1).float()), rel2


def itm_loss(relation_scores, pos_labels, sample_n, multi_w, bin_w):
    """XTrainer.pretrain_under_step ITM loss, M3P/src/xtrainer.py:2357-2372:
    CE over groups of sample_n scores + BCE-with-logits against the one-hot of the
    positive index, weighted by multi_cls_loss_weight / bin_cls_loss_weight."""
    onehot = torch.eye(sample_n, dtype=torch.float32)[pos_labels].reshape(-1)
    ce = F.cross_entropy(relation_scores.view(-1, sample_n).float(), pos_labels)
    bce = F.binary_cross_entropy_with_logits(relation_scores.view(-1).float(), onehot)
    return multi_w * ce + bin_w * bce


def pretrain_losses(sd, n_layers, n_heads, batch, R, sample_n=2, multi_w=0.0, bin_w=1.0,
                    with_itm=True, dropout=0.0, attention_dropout=0.0, keeps=None, with_mrm=False, with_mrfr=False,
                    refine_layers=0, refine_dropout=0.0):
    """The loss half of XTrainer.pretrain_under_step (xtrainer.py:2281-2375) for the
    MLM (+MRM, +MRFR, +ITM) objective: jointfwd -> text slice out[R:] -> MLM CE; image slice out[:R],
    batch-major -> masked-region classification / feature regression; whole sequence, batch-major ->
    relation scores -> ITM loss; total = sum (lambdas = 1)."""
    out = jointfwd(sd, n_layers, n_heads, batch['x'], batch['lengths'], batch['x_img'],
                   batch['lengths_img'], batch['image_loc'], dropout, attention_dropout, keeps,
                   refine_layers=refine_layers, refine_dropout=refine_dropout)
    res = {'out': out}
    total = 0
    if batch['pred_mask'].any():
        _, mlm = predict_mlm(sd, out[R:], batch['pred_mask'], batch['y'])
        res['mlm'] = mlm
        total = total + mlm
    img_out = out[:R].transpose(0, 1)          # (B, R, d), xtrainer.py:2288-2289
    if with_mrm:
        _, mrm = predict_obj(sd, img_out, batch['obj_labels'].reshape(-1))
        res['mrm'] = mrm
        total = total + mrm
    if with_mrfr:
        mrfr = mrfr_loss(predict_mrfr(sd, img_out), batch['obj_labels'], batch['ori_att_feats'])
        res['mrfr'] = mrfr
        total = total + mrfr
    if with_itm:
        rel = predict_relation(sd, out.transpose(0, 1))
        itm = itm_loss(rel, batch['pos_labels'], sample_n, multi_w, bin_w)
        res['rel_scores'] = rel
        res['itm'] = itm
        total = total + itm
    res['total'] = total
    return res


# ----------------------------------------------------------------------------
# optimizer: Adam / AdamInverseSqrtWithWarmup / clip
# ----------------------------------------------------------------------------

def inverse_sqrt_lr(num_updates, lr=1e-4, warmup_updates=4000, warmup_init_lr=1e-7, exp_factor=0.5):
    """AdamInverseSqrtWithWarmup.get_lr_for_step, M3P/src/optim.py:129-133."""
    if num_updates < warmup_updates:
        return warmup_init_lr + num_updates * (lr - warmup_init_lr) / warmup_updates
    return lr * warmup_updates ** exp_factor * num_updates ** -exp_factor


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ as called at xtrainer.py:225: global L2 norm
    over all grads, coef = max_norm / (norm + 1e-6) clamped to 1."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return [g * coef for g in grads], total


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.98, eps=1e-8, weight_decay=0.0):
    """Adam.step, M3P/src/optim.py:45-86 (no amsgrad; eps added to sqrt(v) *before*
    bias correction is folded into the step size; decoupled decay p -= wd*lr*p):
        m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2
        p -= lr * sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)
    ``step`` is the 1-based update count.  Returns new (p, m, v)."""
    m = m * beta1 + (1 - beta1) * g
    v = v * beta2 + (1 - beta2) * g * g
    denom = v.sqrt() + eps
    step_size = lr * math.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
    if weight_decay != 0:
        p = p - weight_decay * lr * p
    p = p - step_size * m / denom
    return p, m, v


class AdamInvSqrt:
    """Stateful wrapper restating AdamInverseSqrtWithWarmup (optim.py:89-139): the
    optimizer is *constructed* with lr = warmup_init_lr (:103-109), each step uses the
    current lr and then sets lr for the next one (:135-139)."""

    def __init__(self, params, lr=1e-4, beta1=0.9, beta2=0.98, eps=1e-8, weight_decay=0.0,
                 warmup_updates=4000, warmup_init_lr=1e-7):
        self.p = [t.clone() for t in params]
        self.m = [torch.zeros_like(t) for t in params]
        self.v = [torch.zeros_like(t) for t in params]
        self.cfg = dict(lr=lr, warmup_updates=warmup_updates, warmup_init_lr=warmup_init_lr)
        self.b1, self.b2, self.eps, self.wd = beta1, beta2, eps, weight_decay
        self.num_updates = 0
        self.lr = warmup_init_lr
        self.steps = [0] * len(params)

    def step(self, grads):
        for i, g in enumerate(grads):
            if g is None:  # optim.py:55-56: params without grad are skipped (no step count)
                continue
            self.steps[i] += 1
            self.p[i], self.m[i], self.v[i] = adam_step(
                self.p[i], g, self.m[i], self.v[i], self.steps[i], self.lr,
                self.b1, self.b2, self.eps, self.wd)
        self.num_updates += 1
        self.lr = inverse_sqrt_lr(self.num_updates, **self.cfg)


def train_step(sd, names, opt, n_layers, n_heads, batch, R, clip=5.0, **loss_kw):
    """One full reference training step (Trainer.optimize amp==-1 path,
    xtrainer.py:218-228): zero_grad -> backward -> clip_grad_norm_ -> step.
    ``sd`` is rebuilt from ``opt.p`` each call.  Returns (losses dict, grads, grad_norm)."""
    leaves = {n: t.detach().clone().requires_grad_(True) for n, t in zip(names, opt.p)}
    res = pretrain_losses(leaves, n_layers, n_heads, batch, R, **loss_kw)
    grads = torch.autograd.grad(res['total'], [leaves[n] for n in names], allow_unused=True)
    live = [g for g in grads if g is not None]
    clipped, norm = clip_grad_norm(live, clip) if clip > 0 else (live, None)
    it = iter(clipped)
    grads_c = [None if g is None else next(it) for g in grads]
    opt.step(grads_c)
    return res, grads, norm


# ----------------------------------------------------------------------------
# retrieval metric (M3P/src/evaluation/xevaluator.py:1621-1657)
# ----------------------------------------------------------------------------

def recall_at_k(score_matrix, gt_index, ks=(1, 5, 10)):
    """Recall@K of a (n_query, n_candidates) score matrix: a query counts as a hit at K
    if its ground-truth candidate is among the K highest scores."""
    order = torch.argsort(score_matrix, dim=1, descending=True)
    rank = (order == gt_index[:, None]).float().argmax(dim=1)
    return {k: float((rank < k).float().mean()) for k in ks}


def retrieval_recalls(scores, labels):
    """The metric arithmetic of evaluate_image_retrieval, xevaluator.py:1621-1657, as plain loops.
    scores, labels: (n_img, n_cap); labels[i, c] == 1 where caption c describes image i.
    image -> sentence: per image, the FIRST positive among its 10 best captions counts once at every K it falls under;
    sentence -> image: per caption, every positive among its 10 best images counts (no early exit in the reference).
    Returns (t2i_r1, t2i_r5, t2i_r10, i2t_r1, i2t_r5, i2t_r10): t2i divided by n_cap, i2t by n_img."""
    n_img, n_cap = scores.shape
    i2t = [0, 0, 0]
    _, pred = scores.topk(min(10, n_cap), dim=-1)
    for i in range(n_img):
        for j, c in enumerate(pred[i].tolist()):
            if labels[i][c] == 1:
                for slot, k in enumerate((1, 5, 10)):
                    if j < k:
                        i2t[slot] += 1
                break
    t2i = [0, 0, 0]
    st, lt = scores.t(), labels.t()
    _, pred = st.topk(min(10, n_img), dim=-1)
    for c in range(n_cap):
        for j, i in enumerate(pred[c].tolist()):
            if lt[c][i] == 1:
                for slot, k in enumerate((1, 5, 10)):
                    if j < k:
                        t2i[slot] += 1
    return tuple(v / n_cap for v in t2i) + tuple(v / n_img for v in i2t)
