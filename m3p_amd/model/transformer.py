"""MI355X-native drop-in for the hot path of M3P's ``TransformerModel``
(reference: M3P/src/model/transformer.py:610-1515).

Same constructor, same ``model(mode, **kwargs)`` dispatcher, same attribute and
state-dict names (so the released checkpoint loads and ``xtrainer.Trainer`` can drive it),
but ``jointfwd`` / ``predict`` run on hand-written gfx950 kernels through the C ABI in
``include/m3p_hip.h`` — there is no PyTorch fallback for those modes on a GPU.

Memory design (288 GB HBM3E per GPU — spend it):
  * every hot parameter is a *view* into one flat fp32 "master" arena; gradients live in a
    second arena with the same layout (``param.grad`` are views), so the gradient
    all-reduce, the global-norm clip and Adam are each a handful of flat streaming kernels;
  * a bf16 working copy (same layout) feeds the MFMA GEMMs; q/k/v weights are adjacent so
    the projection is ONE [3d, d] GEMM; transposed bf16 copies feed the data-gradient
    GEMMs; both are refreshed by the optimizer kernel, not by a per-step cast pass;
  * activations are bf16 [B*S, d] batch-major; everything a layer's backward needs is
    kept resident (≈1 GB/layer at B=256) instead of being recomputed.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as Fn

N_MAX_POSITIONS = 514  # transformer.py:16

ALIGN = 64  # arena offsets in elements (256 B for fp32, 128 B for bf16)


def Embedding(num_embeddings, embedding_dim, padding_idx=None):
    """transformer.py:21-26: N(0, dim^-0.5) init, pad row zeroed."""
    m = nn.Embedding(num_embeddings, embedding_dim, padding_idx=padding_idx)
    nn.init.normal_(m.weight, mean=0, std=embedding_dim ** -0.5)
    if padding_idx is not None:
        nn.init.constant_(m.weight[padding_idx], 0)
    return m


def get_masks(slen, lengths, causal, k=None):
    """transformer.py:59-78 (kept for callers that import it); the hot path does not sync."""
    bs = lengths.size(0)
    alen = torch.arange(slen, dtype=torch.long, device=lengths.device)
    mask = alen < lengths[:, None]
    if causal:
        attn_mask = alen[None, None, :].repeat(bs, slen, 1) <= alen[None, :, None]
    else:
        attn_mask = mask
    return mask, attn_mask


class _Holder(nn.Module):
    """Parameter container used to reproduce the reference's state-dict names for the
    sub-modules the pre-training hot path never executes."""


def _register_path(root, name, tensor):
    parts = name.split('.')
    mod = root
    for p in parts[:-1]:
        if not hasattr(mod, p):
            mod.add_module(p, _Holder())
        mod = getattr(mod, p)
    mod.register_parameter(parts[-1], nn.Parameter(tensor))


def _linear_init(out_f, in_f):
    w = torch.empty(out_f, in_f)
    nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    bound = 1 / math.sqrt(in_f)
    b = torch.empty(out_f).uniform_(-bound, bound)
    return w, b


def cold_param_specs(dim, n_layers, refine_layers):
    """(name, kind, shape...) of the reference's parameters that the MLM+ITM pre-training
    step never touches (enumerated from the reference's state dict: AoA refiner
    transformer.py:274-422, CrossAlignMatrix :425-473, VAE/latent :500-543, cross
    attention :673-698, CLCM / MRFR / object heads :711-727)."""
    specs = [('image_embeddings.image_distbution_embeddings', 'linear', dim, 1600)]
    for i in range(refine_layers):
        p = 'refine_embeddings.layers.%d.' % i
        for j in range(3):
            specs.append((p + 'self_attn.linears.%d' % j, 'linear', dim, dim))
        specs.append((p + 'self_attn.aoa_layer.0', 'linear', 2 * dim, 2 * dim))
        specs.append((p + 'feed_forward.lin1', 'linear', 4 * dim, dim))
        specs.append((p + 'feed_forward.lin2', 'linear', dim, 4 * dim))
        specs.append((p + 'sublayer.0.norm', 'ln', dim))
        specs.append((p + 'sublayer.1.norm', 'ln', dim))
    specs.append(('refine_embeddings.norm', 'ln', dim))
    for n in ('att_weight_c', 'att_weight_q', 'att_weight_cq'):
        specs.append(('cross_alignment.' + n, 'linear', 1, dim))
    specs.append(('cross_alignment.align_output', 'linear', dim, dim))
    specs.append(('cross_alignment.layer_norm', 'ln', dim))
    for i in range(n_layers):
        specs.append(('layer_norm15.%d' % i, 'ln', dim))
    for i in range(n_layers):
        for lin in ('q_lin', 'k_lin', 'v_lin', 'out_lin'):
            specs.append(('encoder_attn.%d.%s' % (i, lin), 'linear', dim, dim))
    for i in range(2):
        specs.append(('latent_transforms.%d.x_to_mu' % i, 'linear', dim, dim))
        specs.append(('latent_transforms.%d.x_to_logvar' % i, 'linear', dim, dim))
        specs.append(('latent_transforms.%d.out_dense' % i, 'linear', dim, 2 * dim))
    for i in range(2):
        specs.append(('original_transforms.%d.dense' % i, 'linear', dim, dim))
        specs.append(('original_transforms.%d.dense_mu' % i, 'linear', dim, dim))
        specs.append(('original_transforms.%d.LayerNorm' % i, 'ln', dim))
    specs.append(('pooled_layer2.dense', 'linear', dim, dim))
    specs.append(('seq_relationship2', 'linear', 1, dim))
    specs.append(('mrfr_dense', 'linear', 2048, dim))
    specs.append(('transformer_obj.dense', 'linear', dim, dim))
    specs.append(('transformer_obj.LayerNorm', 'ln', dim))
    specs.append(('pred_obj_layer.proj', 'linear', 1600, dim))
    return specs


class PredLayer(nn.Module):
    """transformer.py:81-124 (cross-entropy branch; adaptive softmax is out of scope)."""

    def __init__(self, params):
        super().__init__()
        assert params.asm is False, 'adaptive softmax (asm) is outside the MI355X hot path'
        self.asm = params.asm
        self.n_words = params.n_words
        self.pad_index = params.pad_index
        self.proj = nn.Linear(params.emb_dim, params.n_words, bias=True)

    def get_scores(self, x):
        """transformer.py:120-124: word scores of (n, d) hidden states, fp32 (inference; the vocabulary GEMM of the
        training path).  Needs the owning model for the tied bf16 matrix: set by TransformerModel."""
        from ..decoder import word_scores
        return word_scores(self._owner, x)


class BertImageEmbeddings(nn.Module):
    """transformer.py:231-269 parameter holder (forward is fused into the assembly kernel)."""

    def __init__(self, hidden_size):
        super().__init__()
        self.image_embeddings = nn.Linear(2048, hidden_size)
        self.image_location_embeddings = nn.Linear(5, hidden_size)
        self.LayerNorm = nn.LayerNorm(hidden_size, eps=1e-12)


class _AttentionParams(nn.Module):
    """MultiHeadAttention (transformer.py:127-147) parameter holder."""

    def __init__(self, dim):
        super().__init__()
        self.q_lin = nn.Linear(dim, dim)
        self.k_lin = nn.Linear(dim, dim)
        self.v_lin = nn.Linear(dim, dim)
        self.out_lin = nn.Linear(dim, dim)


class _FFNParams(nn.Module):
    """TransformerFFN (transformer.py:213-221) parameter holder."""

    def __init__(self, dim, hidden):
        super().__init__()
        self.lin1 = nn.Linear(dim, hidden)
        self.lin2 = nn.Linear(hidden, dim)


class _Pooler(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dense = nn.Linear(dim, dim)


class TransformerModel(nn.Module):
    ATTRIBUTES = ['encoder', 'with_output', 'eos_index', 'pad_index', 'n_langs', 'n_words', 'dim', 'n_layers',
                  'n_heads', 'hidden_dim', 'dropout', 'attention_dropout', 'asm', 'asm_cutoffs', 'asm_div_value']

    def __init__(self, params, is_encoder, with_output, is_crossModal=False):
        """Same signature and ``params`` fields as transformer.py:614-729."""
        super().__init__()
        self.is_encoder = is_encoder
        self.is_decoder = not is_encoder
        self.with_output = with_output
        self.is_crossModal = is_crossModal
        assert is_crossModal, 'the reference itself requires is_crossModal=True (transformer.py:673-698)'
        # is_encoder=False: the causal decoder of the captioning / translation tasks (same parameter set, n_dec_layers
        # layers, transformer.py:657).  Its inference path (crossfwd(causal=True, src_enc=...), generate, generate_beam)
        # is m3p_amd/decoder.py; its training step is not built.

        self.n_langs = params.n_langs
        self.n_words = params.n_words
        self.eos_index = params.eos_index
        self.pad_index = params.pad_index
        self.id2lang = params.id2lang
        self.lang2id = params.lang2id
        self.english_only = not (self.n_langs > 1)
        assert len(self.id2lang) == len(self.lang2id) == self.n_langs

        self.dim = params.emb_dim
        self.hidden_dim = self.dim * 4
        self.n_heads = params.n_heads
        self.n_layers = params.n_layers if is_encoder else params.n_dec_layers
        assert self.n_layers >= 1
        self.dropout = params.dropout
        self.attention_dropout = params.attention_dropout
        assert self.dim % self.n_heads == 0, 'transformer dim must be a multiple of n_heads'
        assert self.dim // self.n_heads in (32, 64), 'attention kernels are built for head dims 32 and 64'
        assert self.dim % 64 == 0
        assert not params.sinusoidal_embeddings, 'sinusoidal positions are outside the hot path'
        assert params.gelu_activation, 'the fused FFN epilogue is bias+GELU(erf)'
        self.n_refine_layers = int(params.refine_layers)
        # the reference builds its refiner with the constructor defaults dropout=0.1 (transformer.py:288,411,662),
        # whatever params.dropout says
        self.refine_dropout = 0.1
        self.attention_setting = params.attention_setting
        self.use_externel_att = params.use_externel_att

        d = self.dim
        self.position_embeddings = Embedding(N_MAX_POSITIONS, d)
        if params.n_langs > 1:
            self.cross_lang_embeddings = Embedding(self.n_langs, d)   # unused by jointfwd (:937-938)
        self.embeddings = Embedding(self.n_words, d, padding_idx=self.pad_index)
        self.layer_norm_emb = nn.LayerNorm(d, eps=1e-12)
        self.image_embeddings = BertImageEmbeddings(d)

        self.attentions = nn.ModuleList()
        self.layer_norm1 = nn.ModuleList()
        self.ffns = nn.ModuleList()
        self.layer_norm2 = nn.ModuleList()
        for _ in range(self.n_layers):
            self.attentions.append(_AttentionParams(d))
            self.layer_norm1.append(nn.LayerNorm(d, eps=1e-12))
            self.ffns.append(_FFNParams(d, self.hidden_dim))
            self.layer_norm2.append(nn.LayerNorm(d, eps=1e-12))
        self.pooled_layer = _Pooler(d)
        self.seq_relationship = nn.Linear(d, 1)

        # reference parameters that exist but are never executed on this path
        for spec in cold_param_specs(d, self.n_layers, params.refine_layers):
            name, kind = spec[0], spec[1]
            if kind == 'linear':
                w, b = _linear_init(spec[2], spec[3])
                _register_path(self, name + '.weight', w)
                _register_path(self, name + '.bias', b)
            else:
                _register_path(self, name + '.weight', torch.ones(spec[2]))
                _register_path(self, name + '.bias', torch.zeros(spec[2]))

        if self.with_output:
            self.pred_layer = PredLayer(params)
            object.__setattr__(self.pred_layer, '_owner', self)      # (a plain attribute: no cycle in the module tree)
            if params.share_inout_emb:
                self.pred_layer.proj.weight = self.embeddings.weight   # transformer.py:728-729
        self.share_inout_emb = bool(params.share_inout_emb)
        assert self.share_inout_emb, 'the fused MLM head assumes the tied projection (share_inout_emb)'

        # fp8 GEMMs for the encoder layers' projections (BASELINE.json configs[3]; not a reference flag: m3p_amd/fp8.py)
        self.fp8 = bool(getattr(params, 'fp8_gemm', False))
        self._fp8_state = None
        # the encoder-attention sub-layer (encoder_attn / layer_norm15, transformer.py:673-698, executed only by
        # crossfwd(causal=True, src_enc=...)) is trained - i.e. lives in the arena - when the model is a decoder or the
        # run lists translation / auto-encoding steps (train_x.py:215-218); otherwise its parameters stay outside, as
        # never-executed state, and cost the gradient buckets nothing
        self.cross_attention_hot = bool(self.is_decoder or getattr(params, 'mt_steps', None) or getattr(params, 'ae_steps', None)
                                        or getattr(params, 'cross_modal_steps', None)
                                        or getattr(params, 'train_cross_attention', False))
        self._arena = None
        self._cold_w16 = None
        self.base_seed = 0x5EED
        self._fwd_counter = 0
        self.ddp_hook = None   # set by m3p_amd.distributed.DataParallel

    # ------------------------------------------------------------------ arenas
    def hot_named_parameters(self):
        """Hot parameters in arena order; q/k/v weights (and biases) adjacent per layer."""
        out = OrderedDict()
        out['embeddings.weight'] = self.embeddings.weight
        out['pred_layer.proj.bias'] = self.pred_layer.proj.bias
        out['position_embeddings.weight'] = self.position_embeddings.weight
        if self.n_langs > 1:      # language embeddings of the text streams (transformer.py:657, :1059-1060)
            out['cross_lang_embeddings.weight'] = self.cross_lang_embeddings.weight
        out['layer_norm_emb.weight'] = self.layer_norm_emb.weight
        out['layer_norm_emb.bias'] = self.layer_norm_emb.bias
        ie = self.image_embeddings
        out['image_embeddings.image_embeddings.weight'] = ie.image_embeddings.weight
        out['image_embeddings.image_embeddings.bias'] = ie.image_embeddings.bias
        out['image_embeddings.image_location_embeddings.weight'] = ie.image_location_embeddings.weight
        out['image_embeddings.image_location_embeddings.bias'] = ie.image_location_embeddings.bias
        out['image_embeddings.LayerNorm.weight'] = ie.LayerNorm.weight
        out['image_embeddings.LayerNorm.bias'] = ie.LayerNorm.bias
        # AoA refiner (jointfwd(refine_image=True)): inside the embedding gradient bucket - its
        # gradients are the last ones backward produces; the three input projections adjacent like q/k/v
        own = dict(self.named_parameters())
        for i in range(self.n_refine_layers):
            pre = 'refine_embeddings.layers.%d.' % i
            names = [pre + 'self_attn.linears.%d.weight' % j for j in range(3)] + \
                    [pre + 'self_attn.linears.%d.bias' % j for j in range(3)]
            for sub in ('self_attn.aoa_layer.0', 'feed_forward.lin1', 'feed_forward.lin2', 'sublayer.0.norm', 'sublayer.1.norm'):
                names += [pre + sub + '.weight', pre + sub + '.bias']
            for n in names:
                out[n] = own[n]
        if self.n_refine_layers:
            out['refine_embeddings.norm.weight'] = own['refine_embeddings.norm.weight']
            out['refine_embeddings.norm.bias'] = own['refine_embeddings.norm.bias']
        for i in range(self.n_layers):
            a, f = self.attentions[i], self.ffns[i]
            for lin in ('q_lin', 'k_lin', 'v_lin'):
                out['attentions.%d.%s.weight' % (i, lin)] = getattr(a, lin).weight
            for lin in ('q_lin', 'k_lin', 'v_lin'):
                out['attentions.%d.%s.bias' % (i, lin)] = getattr(a, lin).bias
            out['attentions.%d.out_lin.weight' % i] = a.out_lin.weight
            out['attentions.%d.out_lin.bias' % i] = a.out_lin.bias
            out['layer_norm1.%d.weight' % i] = self.layer_norm1[i].weight
            out['layer_norm1.%d.bias' % i] = self.layer_norm1[i].bias
            if self.cross_attention_hot:       # q / k / v adjacent again (one [2d, d] view is the fused key-value projection)
                pre = 'encoder_attn.%d.' % i
                for suffix in ('weight', 'bias'):
                    for lin in ('q_lin', 'k_lin', 'v_lin'):
                        out[pre + lin + '.' + suffix] = own[pre + lin + '.' + suffix]
                for n in (pre + 'out_lin.weight', pre + 'out_lin.bias', 'layer_norm15.%d.weight' % i, 'layer_norm15.%d.bias' % i):
                    out[n] = own[n]
            out['ffns.%d.lin1.weight' % i] = f.lin1.weight
            out['ffns.%d.lin1.bias' % i] = f.lin1.bias
            out['ffns.%d.lin2.weight' % i] = f.lin2.weight
            out['ffns.%d.lin2.bias' % i] = f.lin2.bias
            out['layer_norm2.%d.weight' % i] = self.layer_norm2[i].weight
            out['layer_norm2.%d.bias' % i] = self.layer_norm2[i].bias
        out['pooled_layer.dense.weight'] = self.pooled_layer.dense.weight
        out['pooled_layer.dense.bias'] = self.pooled_layer.dense.bias
        out['seq_relationship.weight'] = self.seq_relationship.weight
        out['seq_relationship.bias'] = self.seq_relationship.bias
        # masked-region heads: trained only when cross_mrm_steps / cross_mrfr_steps are set,
        # otherwise never touched (Adam and the clip norm skip untouched ranges)
        for name in ('transformer_obj.dense.weight', 'transformer_obj.dense.bias', 'transformer_obj.LayerNorm.weight',
                     'transformer_obj.LayerNorm.bias', 'pred_obj_layer.proj.weight', 'pred_obj_layer.proj.bias',
                     'mrfr_dense.weight', 'mrfr_dense.bias', 'pooled_layer2.dense.weight', 'pooled_layer2.dense.bias',
                     'seq_relationship2.weight', 'seq_relationship2.bias'):
            out[name] = own[name]
        return out

    def _apply(self, fn, *a, **kw):
        super()._apply(fn, *a, **kw)
        self._arena = None
        if self.embeddings.weight.is_cuda:
            self._arena = Fn.Arena(self)
        return self

    def decoder_cold_weights(self):
        """bf16 copies of the encoder-attention sub-layer's weights (decoder inference, m3p_amd/decoder.py): views of the
        arena's working copy when the sub-layer is trained, copies made on demand otherwise."""
        if self.cross_attention_hot:
            return Fn.ArenaCrossWeights(self.arena())
        if self._cold_w16 is None:
            from ..decoder import _ColdWeights
            self._cold_w16 = _ColdWeights(self)
        return self._cold_w16.refresh()

    def fp8_state(self):
        if self._fp8_state is None or self._fp8_state.scale.device != self.arena().device:
            from ..fp8 import Fp8State
            self._fp8_state = Fp8State(self.n_layers, self.arena().device)
        return self._fp8_state

    def arena(self):
        if self._arena is None:
            if not self.embeddings.weight.is_cuda:
                raise RuntimeError('m3p_amd.TransformerModel runs on an MI355X only: call .cuda() first '
                                   '(there is no CPU / eager fallback for the hot path)')
            self._arena = Fn.Arena(self)
        return self._arena

    def state_dict(self, *args, **kw):
        # under sharded data parallelism the fp32 master is completed by all-gathers on a side stream: wait for them
        # (stream-level) before anything copies the parameters out
        if self.ddp_hook is not None:
            self.ddp_hook.params_ready(None)
            if getattr(self.ddp_hook, 'master_partial', False):
                # (a collective cannot hide inside state_dict(): often only the master rank calls it)
                raise RuntimeError('the fp32 master of the big matrices is sharded across the data-parallel ranks: call '
                                   'DataParallel.materialize_master() on EVERY rank before state_dict() '
                                   '(Trainer.save_* / end_epoch do)')
        return super().state_dict(*args, **kw)

    def load_state_dict(self, state_dict, strict=True, **kw):
        res = super().load_state_dict(state_dict, strict=strict, **kw)
        if self._arena is not None:
            self._arena.mark_master_changed()
        return res

    # ------------------------------------------------------------------ dispatcher
    def forward(self, mode, **kwargs):
        """transformer.py:731-751."""
        if mode == 'jointfwd':
            return self.jointfwd(**kwargs)
        elif mode == 'predict':
            return self.predict(**kwargs)
        elif mode == 'crossfwd':
            return self.crossfwd(**kwargs)
        elif mode == 'transform':
            return kwargs['tensor']   # transform_original is the identity (transformer.py:1176-1181)
        elif mode in ('fwd', 'ImageEmbed', 'GAN'):
            raise NotImplementedError("mode '%s' is outside the MI355X hot path (SURVEY.md §8f)" % mode)
        raise Exception('Unknown mode: %s' % mode)

    def _next_seed_step(self):
        self._fwd_counter += 1
        return self._fwd_counter

    def jointfwd(self, x, lengths, x_img, lengths_img, causal=False, positions=None, langs=None,
                 image_loc=None, refine_image=False, is_latent=False, text_embed=None):
        """transformer.py:878-968.  x (T,B) int64, x_img (R,B,2048), image_loc (R,B,5) ->
        (S=R+T, B, d) (a transposed view of the batch-major activation, like the reference)."""
        assert not causal and not is_latent, 'causal / is_latent are outside the MI355X hot path'
        T, B = x.size()
        assert lengths.size(0) == B
        R = x_img.size(0)
        p = self.dropout if self.training else 0.0
        pa = self.attention_dropout if self.training else 0.0
        p_ref = None
        if refine_image:     # AoA refiner on the image rows (transformer.py:905-906)
            assert self.n_refine_layers > 0, 'refine_image=True needs params.refine_layers > 0'
            p_ref = self.refine_dropout if self.training else 0.0
        out = self._tag_pass(Fn.EncoderFn.apply(self.layer_norm_emb.weight, self, x, lengths, x_img, lengths_img, image_loc, p, pa,
                                                self._next_seed_step(), p_ref, torch.is_grad_enabled(), text_embed))
        return out.view(B, R + T, self.dim).transpose(0, 1)

    def _tag_pass(self, out):
        """Attach the encoder pass's gradient sink (functional.GradSink) to its output: the heads find it through the
        views the trainer slices off this tensor."""
        out._m3p_sink = getattr(self, '_pending_sink', None)
        self._pending_sink = None
        return out

    @staticmethod
    def _drop_masked_source(src_enc, src_len, enc_mask):
        """``enc_mask`` of crossfwd (transformer.py:1016-1017: ``src_mask &= enc_mask``, the MASS step's hidden source words):
        attention over a key set does not depend on the keys' order, so instead of a second mask in the kernels the allowed
        source rows of every sentence move to the front (a differentiable gather) and ``src_len`` becomes their count."""
        B, S = src_enc.size(0), src_enc.size(1)
        dev = src_enc.device
        ok = enc_mask.to(dev)[:, :S].bool() & (torch.arange(S, device=dev)[None, :] < src_len.to(dev)[:, None])
        order = torch.sort((~ok).to(torch.int8), dim=1, stable=True).indices          # allowed rows first, in their order
        n = ok.sum(dim=1)
        packed = torch.gather(src_enc, 1, order[:, :, None].expand(-1, -1, src_enc.size(2)))
        packed = packed * (torch.arange(S, device=dev)[None, :] < n[:, None])[:, :, None].to(packed.dtype)
        return packed, n

    def crossfwd(self, x, lengths, causal, stream_='text', src_enc=None, src_len=None, positions=None, langs=None,
                 cache=None, enc_mask=None, image_loc=None, **kw):
        """Text-only stream of transformer.py:970-1114 (the mlm_step caller, xtrainer.py:757)."""
        if stream_ == 'img':
            # the image-only encoder pass of the captioning step (:1044-1052): x (R, B, 2048) region features
            assert not causal and src_enc is None and cache is None and positions is None
            assert image_loc is not None and kw.get('image_dist') is None, \
                'the image stream runs without the class-distribution embedding'
            R, B = x.size(0), x.size(1)
            if langs is not None:
                assert self.n_langs > 1 and langs.size() == (R, B)
            p = self.dropout if self.training else 0.0
            pa = self.attention_dropout if self.training else 0.0
            p_ref = None
            if kw.get('refine_image', False):       # the AoA refiner on the stream's rows (:1064-1066)
                assert self.n_refine_layers > 0, 'refine_image=True needs params.refine_layers > 0'
                p_ref = self.refine_dropout if self.training else 0.0
            step = self._next_seed_step()
            h0 = Fn.ImageStreamFn.apply(self.layer_norm_emb.weight, self, x, lengths, image_loc, langs, p, step, p_ref,
                                        torch.is_grad_enabled())
            out = self._tag_pass(Fn.EncoderFn.apply(self.layer_norm_emb.weight, self, None, lengths, None, None, None, p, pa, step,
                                                    None, torch.is_grad_enabled(), None, None, h0))
            return out.view(B, R, self.dim).transpose(0, 1)
        assert stream_ == 'text'
        if causal:       # the decoder: causal self-attention (+ attention over src_enc), key / value cache (:1011-1091)
            if enc_mask is not None:
                src_enc, src_len = self._drop_masked_source(src_enc, src_len, enc_mask)
            if torch.is_grad_enabled() and self.training:
                # teacher-forced training pass (mt_step / ae_step, xtrainer.py:1383-1441): the whole target at once
                assert cache is None
                T, B = x.size()
                out = Fn.DecoderFn.apply(self.layer_norm_emb.weight, self, x, lengths, src_enc, src_len, langs, self.dropout,
                                         self.attention_dropout, self._next_seed_step(), positions, kw.get('text_embed'))
                return out.view(B, T, self.dim).transpose(0, 1)
            from .. import decoder
            return decoder.decoder_forward(self, x, lengths, src_enc=src_enc, src_len=src_len, positions=positions,
                                           langs=langs, cache=cache)
        assert src_enc is None and cache is None, 'the non-causal text stream takes no source encoding or cache (the mlm_step caller)'
        T, B = x.size()
        if positions is not None:  # transformer.py:1057-1058 (TLM batches: positions restart at the second sentence)
            assert positions.size() == (T, B)
        if langs is not None:      # transformer.py:1059-1060: + cross_lang_embeddings(langs) on the text rows
            assert self.n_langs > 1 and langs.size() == (T, B), 'language ids need a model with n_langs > 1'
        p = self.dropout if self.training else 0.0
        pa = self.attention_dropout if self.training else 0.0
        out = self._tag_pass(Fn.EncoderFn.apply(self.layer_norm_emb.weight, self, x, lengths, None, None, None, p, pa,
                                                self._next_seed_step(), None, torch.is_grad_enabled(), None, langs, None,
                                                positions))
        return out.view(B, T, self.dim).transpose(0, 1)

    def predict(self, tensor, pred_mask=None, y=None, get_scores=None, is_obj=False, is_relation=False,
                is_mrfr=False, is_clcm=False):
        """transformer.py:1183-1214."""
        if is_relation or is_clcm:
            # BertPooler (:546-558) + seq_relationship (:1194-1197), or the second pair for the CLCM pass
            # (:1198-1201): GEMMs + csrc/itm.hip; the score stays on the device (the reference moves it to the CPU)
            first = tensor[:, 0]
            if first.dtype != Fn.BF16:
                first = first.to(Fn.BF16)
            if first.stride(-1) != 1:
                first = first.contiguous()
            sink = Fn.first_rows_sink(self, tensor)      # (GradSink of the pass, its row buffer, rows of tensor[:, 0]) or Nones
            if is_clcm:
                return Fn.ItmHeadFn.apply(first, self, 'pooled_layer2', 'seq_relationship2', *sink)
            return Fn.ItmHeadFn.apply(first, self, 'pooled_layer', 'seq_relationship', *sink)
        if is_obj:
            # transformer.py:1205-1210: (scores, loss) of the masked-region classification head; scores are
            # not materialised for all B*R rows (only the masked rows enter the ignore_index mean)
            return None, Fn.mrm_head(self, tensor, y)
        if is_mrfr:
            # transformer.py:1202-1204: the bare regression mrfr_dense(tensor) of every row handed in (the fused
            # masked loss XTrainer uses is m3p_amd.functional.mrfr_head)
            return Fn.mrfr_dense_rows(self, tensor)
        loss, scores = Fn.mlm_head(self, tensor, pred_mask, y, bool(get_scores))
        return scores, loss

    def generate(self, src_enc, src_len, tgt_lang_id, max_len=200, sample_temperature=None, cross_modal=True):
        """transformer.py:1216-1317 (greedy / sampled decoding)."""
        from .. import decoder
        return decoder.generate(self, src_enc, src_len, tgt_lang_id, max_len=max_len, sample_temperature=sample_temperature)

    def generate_beam(self, src_enc, src_len, tgt_lang_id, beam_size, length_penalty, early_stopping, max_len=200):
        """transformer.py:1319-1515 (beam search)."""
        from .. import decoder
        return decoder.generate_beam(self, src_enc, src_len, tgt_lang_id, beam_size, length_penalty, early_stopping,
                                     max_len=max_len)
