"""Deterministic synthetic inputs and weights for the M3P pre-training hot path.

Shapes / dtypes / normalisation follow the reference's data contract:
  * ``x``        (T, B) int64 token ids, BOS=0 first, EOS=2 last valid, PAD=1 after
                 (reference collate ``batch_sentences_v2``, M3P/src/xtrainer.py:855-880)
  * ``x_img``    (R, B, 2048) fp32 region features, L2-normalised per region
                 (M3P/src/data/dataset_pretrain.py:326,379)
  * ``image_loc``(R, B, 5) fp32 box geometry, L2-normalised (dataset_pretrain.py:294-301)
  * MLM labels   (T, B) int64, -1 = not predicted (xtrainer.py:2226-2232)

Everything is drawn from ``numpy.random.RandomState`` streams, which are frozen
across NumPy versions, so fixtures generated in one container reproduce in another.
Used by the golden-vector generator, the tests and ``bench.py``.
"""
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np
import torch

N_MAX_POSITIONS = 514  # M3P/src/model/transformer.py:16

BOS, PAD, EOS = 0, 1, 2


def model_params(emb_dim, n_heads, n_layers, n_words, dropout=0.0, attention_dropout=0.0,
                 refine_layers=0, **extra):
    """The flat ``params`` Namespace fields the TransformerModel ctor reads
    (M3P/src/model/transformer.py:627-679, 725-729; PredLayer :88-102)."""
    p = SimpleNamespace(
        n_langs=1, n_words=n_words, eos_index=EOS, pad_index=PAD, mask_index=n_words - 1,
        id2lang={0: 'en'}, lang2id={'en': 0},
        emb_dim=emb_dim, n_heads=n_heads, n_layers=n_layers, n_dec_layers=-1,
        dropout=dropout, attention_dropout=attention_dropout,
        sinusoidal_embeddings=False, refine_layers=refine_layers,
        attention_setting='v1', use_externel_att=False, gelu_activation=True,
        asm=False, share_inout_emb=True,
    )
    for k, v in extra.items():
        setattr(p, k, v)
    return p


CONFIGS = {
    # BASELINE.json configs[0]: the reference's CPU-runnable case
    'cfg1': dict(emb_dim=128, n_heads=4, n_layers=2, n_words=1000, T=64, R=10, B=8, n_pred=9),
    # BASELINE.json configs[1]: M3P-base on one MI355X
    'cfg2': dict(emb_dim=768, n_heads=12, n_layers=12, n_words=250002, T=128, R=36, B=256, n_pred=19),
    # configs[2]: the same model data-parallel over 8 MI355X, global batch 8192 = 1024 sequences per GPU (B is the per-GPU share)
    'cfg3': dict(emb_dim=768, n_heads=12, n_layers=12, n_words=250002, T=128, R=36, B=1024, n_pred=19),
    # configs[3]: M3P-large
    'cfg4': dict(emb_dim=1024, n_heads=16, n_layers=24, n_words=250002, T=256, R=100, B=64, n_pred=38),
    # configs[4]: ITM fine-tune shape
    'cfg5': dict(emb_dim=768, n_heads=12, n_layers=12, n_words=250002, T=80, R=36, B=96, n_pred=0),
}


def hot_param_shapes(p):
    """(name, shape) of every parameter the pre-training step touches, in the
    reference's state-dict naming.  ``pred_layer.proj.weight`` is tied to
    ``embeddings.weight`` (transformer.py:728-729) and therefore not listed twice."""
    d, V, L = p.emb_dim, p.n_words, p.n_layers
    out = OrderedDict()
    out['position_embeddings.weight'] = (N_MAX_POSITIONS, d)
    out['embeddings.weight'] = (V, d)
    if p.n_langs > 1:
        out['cross_lang_embeddings.weight'] = (p.n_langs, d)
    out['layer_norm_emb.weight'] = (d,)
    out['layer_norm_emb.bias'] = (d,)
    out['image_embeddings.image_embeddings.weight'] = (d, 2048)
    out['image_embeddings.image_embeddings.bias'] = (d,)
    out['image_embeddings.image_location_embeddings.weight'] = (d, 5)
    out['image_embeddings.image_location_embeddings.bias'] = (d,)
    out['image_embeddings.LayerNorm.weight'] = (d,)
    out['image_embeddings.LayerNorm.bias'] = (d,)
    for i in range(L):
        for lin in ('q_lin', 'k_lin', 'v_lin', 'out_lin'):
            out['attentions.%d.%s.weight' % (i, lin)] = (d, d)
            out['attentions.%d.%s.bias' % (i, lin)] = (d,)
        out['layer_norm1.%d.weight' % i] = (d,)
        out['layer_norm1.%d.bias' % i] = (d,)
        out['ffns.%d.lin1.weight' % i] = (4 * d, d)
        out['ffns.%d.lin1.bias' % i] = (4 * d,)
        out['ffns.%d.lin2.weight' % i] = (d, 4 * d)
        out['ffns.%d.lin2.bias' % i] = (d,)
        out['layer_norm2.%d.weight' % i] = (d,)
        out['layer_norm2.%d.bias' % i] = (d,)
    out['pooled_layer.dense.weight'] = (d, d)
    out['pooled_layer.dense.bias'] = (d,)
    out['seq_relationship.weight'] = (1, d)
    out['seq_relationship.bias'] = (1,)
    out['pred_layer.proj.bias'] = (V,)
    return out


def decoder_param_shapes(p):
    """(name, shape) of the parameters the causal decoder (is_encoder=False) executes in
    crossfwd(causal=True, src_enc=...) / generate / generate_beam (transformer.py:1050-1102): the text embeddings, per
    layer the self-attention, the encoder attention + layer_norm15, the FFN, and the tied output bias."""
    d, V, L = p.emb_dim, p.n_words, p.n_dec_layers
    out = OrderedDict()
    out['position_embeddings.weight'] = (N_MAX_POSITIONS, d)
    out['embeddings.weight'] = (V, d)
    if p.n_langs > 1:
        out['cross_lang_embeddings.weight'] = (p.n_langs, d)
    out['layer_norm_emb.weight'] = (d,)
    out['layer_norm_emb.bias'] = (d,)
    for i in range(L):
        for blk in ('attentions', 'encoder_attn'):
            for lin in ('q_lin', 'k_lin', 'v_lin', 'out_lin'):
                out['%s.%d.%s.weight' % (blk, i, lin)] = (d, d)
                out['%s.%d.%s.bias' % (blk, i, lin)] = (d,)
        for ln in ('layer_norm1', 'layer_norm15', 'layer_norm2'):
            out['%s.%d.weight' % (ln, i)] = (d,)
            out['%s.%d.bias' % (ln, i)] = (d,)
        out['ffns.%d.lin1.weight' % i] = (4 * d, d)
        out['ffns.%d.lin1.bias' % i] = (4 * d,)
        out['ffns.%d.lin2.weight' % i] = (d, 4 * d)
        out['ffns.%d.lin2.bias' % i] = (d,)
    out['pred_layer.proj.bias'] = (V,)
    return out


DECODER_CASES = {
    # tag: model geometry and search settings of the decoder golden vectors (oracle/gen_goldens.py decoder)
    'mono': dict(emb_dim=128, n_heads=4, n_dec_layers=2, n_words=1000, n_langs=1, bs=5, S=12, T=10, max_len=18,
                 tgt_lang_id=None, beam_size=0, eos_bias=-3.0, eos_ramp=1.0, seed=31),
    'multi': dict(emb_dim=128, n_heads=4, n_dec_layers=2, n_words=1000, n_langs=2, bs=4, S=9, T=8, max_len=14,
                  tgt_lang_id=1, beam_size=3, eos_bias=-2.0, eos_ramp=0.3, seed=47),
    'wide': dict(emb_dim=256, n_heads=4, n_dec_layers=3, n_words=2000, n_langs=2, bs=3, S=20, T=12, max_len=16,
                 tgt_lang_id=0, beam_size=4, eos_bias=1.0, eos_ramp=0.3, seed=59),
}


def decoder_case(tag):
    """params, deterministic weights and inputs of one decoder golden case."""
    c = DECODER_CASES[tag]
    langs = {0: 'en', 1: 'zh'} if c['n_langs'] > 1 else {0: 'en'}
    p = model_params(c['emb_dim'], c['n_heads'], 1, c['n_words'], n_dec_layers=c['n_dec_layers'], n_langs=c['n_langs'],
                     id2lang=langs, lang2id={v: k for k, v in langs.items()})
    sd = golden_state_dict(decoder_param_shapes(p), seed=c['seed'], scale=0.05)
    # so that hypotheses end at different, input-dependent steps: <EOS> starts below the other words and every position
    # embedding leans a little further towards the <EOS> embedding (which is also its output vector: tied matrix)
    sd['pred_layer.proj.bias'][EOS] += c['eos_bias']
    e_hat = sd['embeddings.weight'][EOS] / sd['embeddings.weight'][EOS].norm()
    ramp = torch.arange(64, dtype=torch.float32)[:, None] * c.get('eos_ramp', 0.0)
    sd['position_embeddings.weight'][:64] += ramp * e_hat[None, :]
    rs = np.random.RandomState(c['seed'] + 1)
    src_enc = torch.from_numpy(rs.standard_normal((c['bs'], c['S'], c['emb_dim'])).astype(np.float32))
    src_len = torch.from_numpy(rs.randint(c['S'] // 2, c['S'] + 1, size=c['bs'])).long()
    src_len[0] = c['S']
    x = torch.from_numpy(rs.randint(3, c['n_words'], size=(c['T'], c['bs']))).long()
    lengths = torch.from_numpy(rs.randint(c['T'] // 2, c['T'] + 1, size=c['bs'])).long()
    lengths[-1] = c['T']
    x[0] = EOS
    for b in range(c['bs']):
        x[int(lengths[b]) - 1, b] = EOS
        x[int(lengths[b]):, b] = PAD
    return c, p, sd, src_enc, src_len, x, lengths


def region_head_param_shapes(p):
    """Parameters of the two masked-region objectives: MRM = BertPredictionHeadTransform
    (transformer.py:595-606) + ObjPredLayer (:562-584); MRFR = mrfr_dense (:718).  Kept apart from
    hot_param_shapes so that the golden weight stream of the MLM+ITM goldens does not move."""
    d = p.emb_dim
    out = OrderedDict()
    out['transformer_obj.dense.weight'] = (d, d)
    out['transformer_obj.dense.bias'] = (d,)
    out['transformer_obj.LayerNorm.weight'] = (d,)
    out['transformer_obj.LayerNorm.bias'] = (d,)
    out['pred_obj_layer.proj.weight'] = (1600, d)
    out['pred_obj_layer.proj.bias'] = (1600,)
    out['mrfr_dense.weight'] = (2048, d)
    out['mrfr_dense.bias'] = (2048,)
    return out


def clcm_head_param_shapes(p):
    """Second pooler + relation head of the cross-lingual contrastive pass (transformer.py:715-716, :1198-1201)."""
    d = p.emb_dim
    out = OrderedDict()
    out['pooled_layer2.dense.weight'] = (d, d)
    out['pooled_layer2.dense.bias'] = (d,)
    out['seq_relationship2.weight'] = (1, d)
    out['seq_relationship2.bias'] = (1,)
    return out


def refiner_param_shapes(p):
    """AoA_Refiner_Core parameters (transformer.py:287-422, built at :662 with params.refine_layers layers) in the
    reference's state-dict naming."""
    d = p.emb_dim
    out = OrderedDict()
    for i in range(p.refine_layers):
        pre = 'refine_embeddings.layers.%d.' % i
        for j in range(3):
            out[pre + 'self_attn.linears.%d.weight' % j] = (d, d)
            out[pre + 'self_attn.linears.%d.bias' % j] = (d,)
        out[pre + 'self_attn.aoa_layer.0.weight'] = (2 * d, 2 * d)
        out[pre + 'self_attn.aoa_layer.0.bias'] = (2 * d,)
        out[pre + 'feed_forward.lin1.weight'] = (4 * d, d)
        out[pre + 'feed_forward.lin1.bias'] = (4 * d,)
        out[pre + 'feed_forward.lin2.weight'] = (d, 4 * d)
        out[pre + 'feed_forward.lin2.bias'] = (d,)
        for k in (0, 1):
            out[pre + 'sublayer.%d.norm.weight' % k] = (d,)
            out[pre + 'sublayer.%d.norm.bias' % k] = (d,)
    out['refine_embeddings.norm.weight'] = (d,)
    out['refine_embeddings.norm.bias'] = (d,)
    return out


def make_region_targets(R, B, seed=2468, p_mask=0.3, n_objs=1600):
    """Synthetic MRM / MRFR targets: obj_labels (B, R) int64, -1 = region not masked, else the object class
    of the masked region (xtrainer.py:2263, 2325-2328); ori_att_feats (B, R, 2048) fp32 = the original region
    features the MRFR head regresses (xtrainer.py:2337-2346).  At least one region per batch is masked."""
    rs = np.random.RandomState(seed)
    masked = rs.rand(B, R) < p_mask
    masked[0, 0] = True
    labels = np.where(masked, rs.randint(0, n_objs, size=(B, R)), -1).astype(np.int64)
    feats = rs.standard_normal((B, R, 2048)).astype(np.float32)
    feats /= np.linalg.norm(feats, axis=-1, keepdims=True)
    return dict(obj_labels=torch.from_numpy(labels), ori_att_feats=torch.from_numpy(feats))


def golden_weight(name, shape, rs, scale=0.02):
    """One tensor of the golden weight stream: N(0,1)*scale, LayerNorm gains 1+that
    (SURVEY §8c 'Golden inputs')."""
    w = rs.standard_normal(shape).astype(np.float32) * np.float32(scale)
    is_ln = ('layer_norm' in name or 'LayerNorm' in name)
    if is_ln and name.endswith('weight'):
        w = w + np.float32(1.0)
    return w


def golden_state_dict(names_shapes, seed=1234, scale=0.02, pad_index=PAD):
    """Deterministic weights for every (name, shape) in *sorted key order*; the
    embedding pad row is zeroed like the reference's init (transformer.py:21-26)."""
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for name in sorted(names_shapes):
        w = golden_weight(name, tuple(names_shapes[name]), rs, scale)
        if name == 'embeddings.weight' and pad_index is not None:
            w[pad_index] = 0
        sd[name] = torch.from_numpy(w)
    return sd


def make_batch(T, R, B, n_words, n_pred, seed=5678, ragged=True, sample_n=2):
    """Synthetic pre-training batch.

    Returns a dict of CPU tensors:
      x (T,B) int64 [already masked with <mask>=V-1 at the predicted positions],
      lengths (B,), x_labels (T,B) int64 (-1 = not predicted, else original id),
      pred_mask (T,B) bool, y (n_pred*B,) int64 in (s,b) row order,
      x_img (R,B,2048) fp32, image_loc (R,B,5) fp32, lengths_img (B,),
      pos_labels (B/sample_n,) int64, itm_targets (B,) fp32 one-hot flattened.
    """
    rs = np.random.RandomState(seed)
    mask_index = n_words - 1
    if ragged:
        lengths = rs.randint(T // 2, T + 1, size=B).astype(np.int64)
        lengths[0] = T  # one full-length row so slen == T
    else:
        lengths = np.full(B, T, dtype=np.int64)
    x = np.full((T, B), PAD, dtype=np.int64)
    labels = np.full((T, B), -1, dtype=np.int64)
    for b in range(B):
        n = int(lengths[b])
        x[0, b] = BOS
        x[1:n - 1, b] = rs.randint(4, n_words - 1, size=n - 2)  # ids in [4, V-2]
        x[n - 1, b] = EOS
        if n_pred > 0:
            k = min(n_pred, n - 2)
            pos = 1 + rs.permutation(n - 2)[:k]  # uniform in [1, len-2]
            labels[pos, b] = x[pos, b]
            x[pos, b] = mask_index
    pred_mask = labels != -1
    y = labels[pred_mask]  # boolean gather on (T,B) -> (s,b) row order (transformer.py:1208)

    feat = rs.standard_normal((R, B, 2048)).astype(np.float32)
    feat /= np.linalg.norm(feat, axis=-1, keepdims=True)
    loc = rs.uniform(0.0, 1.0, size=(R, B, 5)).astype(np.float32)
    loc /= np.linalg.norm(loc, axis=-1, keepdims=True)
    lengths_img = np.full(B, R, dtype=np.int64)

    n_groups = max(B // sample_n, 1)
    pos_labels = (np.arange(n_groups) % sample_n).astype(np.int64)
    itm = np.eye(sample_n, dtype=np.float32)[pos_labels].reshape(-1)[:B]

    return dict(
        x=torch.from_numpy(x), lengths=torch.from_numpy(lengths),
        x_labels=torch.from_numpy(labels), pred_mask=torch.from_numpy(pred_mask),
        y=torch.from_numpy(y),
        x_img=torch.from_numpy(feat), image_loc=torch.from_numpy(loc),
        lengths_img=torch.from_numpy(lengths_img),
        pos_labels=torch.from_numpy(pos_labels), itm_targets=torch.from_numpy(itm),
    )


def text_langs_case():
    """Two-language cfg1 model + batch of the language-embedding golden (sentence b is in language b % 2)."""
    cfg = CONFIGS['cfg1']
    P = model_params(cfg['emb_dim'], cfg['n_heads'], cfg['n_layers'], cfg['n_words'], n_langs=2,
                           id2lang={0: 'en', 1: 'zh'}, lang2id={'en': 0, 'zh': 1})
    sd = golden_state_dict(hot_param_shapes(P), seed=4321)
    batch = make_batch(cfg['T'], cfg['R'], cfg['B'], cfg['n_words'], cfg['n_pred'], seed=99)
    langs = (torch.arange(cfg['B']) % 2)[None, :].expand(cfg['T'], cfg['B']).contiguous()
    return cfg, P, sd, batch, langs


def cross_attention_param_shapes(p):
    """(name, shape) of the encoder-attention sub-layer (transformer.py:673-698) of every layer."""
    d, out = p.emb_dim, OrderedDict()
    for i in range(p.n_layers):
        for lin in ('q_lin', 'k_lin', 'v_lin', 'out_lin'):
            out['encoder_attn.%d.%s.weight' % (i, lin)] = (d, d)
            out['encoder_attn.%d.%s.bias' % (i, lin)] = (d,)
        out['layer_norm15.%d.weight' % i] = (d,)
        out['layer_norm15.%d.bias' % i] = (d,)
    return out


def mt_case():
    """Two-language encoder_only model (cfg1 width, 2 layers) with a trained encoder-attention sub-layer, and a translation
    batch (source sentences x1 in language 0, targets x2 in language 1) for the mt_step golden."""
    P = model_params(128, 4, 2, 1000, n_langs=2, id2lang={0: 'en', 1: 'zh'}, lang2id={'en': 0, 'zh': 1},
                     mt_steps=[('en', 'zh')], encoder_only=True)
    shapes = hot_param_shapes(P)
    shapes.update(cross_attention_param_shapes(P))
    sd = golden_state_dict(shapes, seed=777)
    rs = np.random.RandomState(778)

    def sentences(T, B):
        x = torch.from_numpy(rs.randint(3, P.n_words, size=(T, B))).long()
        lengths = torch.from_numpy(rs.randint(T // 2, T + 1, size=B)).long()
        lengths[0] = T
        x[0] = EOS
        for b in range(B):
            x[int(lengths[b]) - 1, b] = EOS
            x[int(lengths[b]):, b] = PAD
        return x, lengths

    x1, len1 = sentences(14, 6)
    x2, len2 = sentences(11, 6)
    return P, sd, x1, len1, x2, len2


def mt_targets(x2, len2):
    """xtrainer.py:1410-1413: predict word t + 1 from position t for every position but a sentence's last."""
    alen = torch.arange(int(len2.max()), dtype=torch.long)
    pred_mask = alen[:, None] < len2[None] - 1
    y = x2[1:].masked_select(pred_mask[:-1])
    return pred_mask, y


def ic_case():
    """The translation model of mt_case with an image source: R = 10 regions per image (ragged), captions x2."""
    P, sd, _, _, x2, len2 = mt_case()
    rs = np.random.RandomState(779)
    R, B = 10, x2.shape[1]
    x_img = torch.from_numpy(rs.standard_normal((R, B, 2048)).astype(np.float32))
    loc = torch.from_numpy(rs.uniform(0, 1, size=(R, B, 5)).astype(np.float32))
    img_len = torch.from_numpy(rs.randint(R // 2, R + 1, size=B)).long()
    img_len[0] = R
    return P, sd, x_img, loc, img_len, x2, len2


def mt_ic_case():
    """The translation model of mt_case with an image beside the source sentence (the multimodal-translation step):
    R = 10 regions per image, all valid - the loaders always hand out full region sets (dataset_pretrain.py:313), which is
    what makes jointfwd's prefix mask (image length + text length) the right one."""
    P, sd, x_src, len_src, x2, len2 = mt_case()
    rs = np.random.RandomState(780)
    R, B = 10, x2.shape[1]
    x_img = torch.from_numpy(rs.standard_normal((R, B, 2048)).astype(np.float32))
    x_img = x_img / x_img.norm(dim=-1, keepdim=True)
    loc = torch.from_numpy(rs.uniform(0, 1, size=(R, B, 5)).astype(np.float32))
    img_len = torch.full((B,), R, dtype=torch.long)
    return P, sd, x_src, len_src, x_img, loc, img_len, x2, len2


def token_stream(seed=41, n_sent=23, V=1000):
    """A monolingual corpus as the reference's binarised files hold it: word ids with an EOS after every sentence, the
    (start, end) position of each sentence (end = its EOS), and a language id per token (for the StreamDataset golden)."""
    rs = np.random.RandomState(seed)
    sents = [rs.randint(4, V - 1, size=rs.randint(1, 12)) for _ in range(n_sent)]
    sent = np.concatenate([np.concatenate([s, [EOS]]) for s in sents]).astype(np.int32)
    ends = np.cumsum([len(s) + 1 for s in sents]) - 1
    pos = np.stack([ends - np.array([len(s) for s in sents]), ends], axis=1).astype(np.int64)
    langs = rs.randint(0, 2, size=len(sent)).astype(np.int32)
    return sent, pos, langs


def ic_refine_case():
    """ic_case with a two-layer AoA refiner on the image stream (crossfwd(stream_='img', refine_image=True)): the captioning
    runs of the reference keep the parser's default refine_image=True (train_x.py:285)."""
    P, sd, x_img, loc, img_len, x2, len2 = ic_case()
    P.refine_layers = 2
    sd = OrderedDict(sd)
    sd.update(golden_state_dict(refiner_param_shapes(P), seed=2468, pad_index=None))
    w = torch.from_numpy(np.random.RandomState(781).standard_normal((x_img.shape[0], x_img.shape[1], P.emb_dim)).astype(np.float32))
    return P, sd, x_img, loc, img_len, w


def trainer_params(**over):
    """The fields Trainer / XTrainer read beside the model's (xtrainer.py:37-136, :734-770), at the values the goldens of
    the text steps are recorded with (dropout 0 models, fp32, no accumulation, clip 5)."""
    p = dict(encoder_only=True, epoch_size=100, stopping_criterion='', amp=-1, fp16=False, accumulate_gradients=1,
             multi_gpu=False, local_rank=0, word_mask=0.8, word_keep=0.1, word_rand=0.1, validation_metrics='',
             dump_path='/nonexistent_m3p_dump', reload_checkpoint='',
             optimizer='adam_inverse_sqrt,beta1=0.9,beta2=0.98,lr=0.0001', use_memory=0, clip_grad_norm=5,
             pc_steps=[], ae_steps=[], mass_steps=[], bt_steps=[], cross_modal_steps=[], cross_rel_steps=[],
             cross_mass_steps=[], cross_ae_steps=[], cross_gan_steps=[], cross_mlm_steps=[], cross_mrm_steps=[],
             cross_mrfr_steps=[], cross_clcm_steps=[], max_region_num=10, sample_n=2, is_latent=False, refine_image=False,
             multi_cls_loss_weight=0, bin_cls_loss_weight=1, batch_size=8, sample_alpha=0, word_pred=0.15, is_ntg=False,
             group_by_size=False, is_freelb=False, t2i_flag=True, i2t_flag=True, langs=['en'])
    for lam in ('lambda_clm', 'lambda_mlm', 'lambda_pc', 'lambda_ae', 'lambda_mt', 'lambda_bt', 'lambda_mass', 'lambda_ic',
                'lambda_imlm', 'lambda_ida', 'lambda_tifg', 'lambda_rel', 'lambda_mrm', 'lambda_mrfr', 'lambda_t2i', 'lambda_i2t'):
        p[lam] = '1'
    p.update(over)
    return p
