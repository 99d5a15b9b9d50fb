"""Trainer surface of the hot path: the parts of ``Trainer`` / ``XTrainer``
(M3P/src/xtrainer.py:35-826, 1128-2402) that ``train_x.py``'s main loop (:432-508) reaches on
the understanding / retrieval tasks, driving the MI355X model.

Entry points kept with the reference's signatures and batch tuples:
``mlm_step(lang1, lang2, lambda)`` (:734-770, with ``generate_batch`` / ``round_batch`` /
``mask_out``), ``pretrain_rel_step`` (:1879-1886), ``rel_step`` (:1867-1877),
``pretrain_under_step`` (:2234-2402), ``t2i_step`` / ``i2t_step`` (:1888-2018), ``optimize``
(:205-243), ``iter`` / ``print_stats`` (:245-289, the reference's own sent/s meter),
``save_model`` / ``save_checkpoint`` / ``reload_checkpoint`` / ``save_periodic`` /
``save_best_model`` / ``end_epoch`` (:511-650), ``get_iterator`` / ``get_batch`` (:1148-1206).

Changed on purpose (SURVEY.md §7 "host side clean"): no per-step host syncs - the NaN check
(:210), the ``.cpu()`` ITM loss (:2367-2370) and the ``loss.item()`` statistics (:2317, :2374)
stay on the device and are only read when ``print_stats`` prints; clip + Adam + zero_grad are
one fused kernel pass; DDP is our bucketed reducer (m3p_amd/distributed.py).  ``mt_step`` (:1383-1441) and ``ic_step`` (:1443-1515) are the
translation / captioning steps on the causal stream.  Not built: the FreeLB / sliding-window steps (out of scope).
"""
import math
import os
import time
from collections import OrderedDict
from logging import getLogger

import numpy as np
import torch
import torch.nn.functional as F

from . import masking
from .collate import (batch_sentences, batch_sentences_v2, caption_collate, mt_caption_collate, ntg_collate,  # noqa: F401
                      retrieval_collate, retrieval_pretrain_collate, slide_collate)
from .distributed import DataParallel
from .optim import get_optimizer
from .utils import concat_batches, parse_lambda_config, to_cuda, update_lambdas

logger = getLogger()

_REGION_HEADS = {
    'mlm': ('embeddings.weight', 'pred_layer.proj.bias'),
    'mrm': ('transformer_obj.dense.weight', 'transformer_obj.dense.bias', 'transformer_obj.LayerNorm.weight',
            'transformer_obj.LayerNorm.bias', 'pred_obj_layer.proj.weight', 'pred_obj_layer.proj.bias'),
    'mrfr': ('mrfr_dense.weight', 'mrfr_dense.bias'),
}


def _add_loss(total, coeff, loss):
    """total + coeff * loss without the launches a unit coefficient / an empty total would cost."""
    term = loss if coeff == 1 else coeff * loss
    return term if total is None else total + term


def _unwrap(model):
    return model.module if isinstance(model, DataParallel) else model


def _stat_names(params):
    """Loss statistics of the steps this build runs (xtrainer.py:101-130 lists them for every task)."""
    g = lambda k: getattr(params, k, [])   # noqa: E731
    names = ['MLM-%s' % l for l in g('langs')] + ['MT-%s-%s' % (l1, l2) for l1, l2 in (g('mt_steps') or [])] + \
        ['AE-%s' % l for l in (g('ae_steps') or [])]
    for key, steps in (('CMLM', g('cross_mlm_steps')), ('MRM', g('cross_mrm_steps')), ('MRFR', g('cross_mrfr_steps')),
                       ('t2i', g('cross_rel_steps')), ('i2t', g('cross_rel_steps'))):
        names += ['%s-%s' % (key, l1) for l1, _ in steps]
    return names


class Trainer(object):
    MODEL_NAMES = ['model']

    def __init__(self, data, params):
        """xtrainer.py:37-136."""
        self.params = params
        self.data = data
        self.epoch_size = params.epoch_size
        if self.epoch_size == -1:
            self.epoch_size = self.data
            assert self.epoch_size > 0
        # early stopping: "metric,patience", a leading underscore = lower is better
        self.stopping_criterion = self.best_stopping_criterion = None
        crit = getattr(params, 'stopping_criterion', '')
        if crit != '':
            metric, _, patience = crit.partition(',')
            assert patience.isdigit(), 'stopping_criterion is "<metric>,<patience>"'
            self.decrease_counts_max, self.decrease_counts = int(patience), 0
            biggest = not metric.startswith('_')
            self.stopping_criterion = (metric if biggest else metric[1:], biggest)
            self.best_stopping_criterion = -1e12 if biggest else 1e12
        self.iterators = {}
        self.set_parameters()
        assert params.amp >= 1 or not params.fp16
        # bf16 compute with fp32 master weights is the only precision mode of this build;
        # `amp`/`fp16` are accepted for flag compatibility (no loss scaling: bf16 keeps fp32's range).
        # The reference's amp == -1 branch (xtrainer.py:218-228) trains in fp32; here it runs the SAME bf16 kernels
        # (MI355X fp32 MFMA is 1/16 of the bf16 rate) - said once per run so nobody reads the flag as a precision promise
        if params.amp == -1 and not getattr(Trainer, '_warned_fp32', False):
            Trainer._warned_fp32 = True
            logger.warning('amp == -1 requests fp32 training in the reference; this build computes in bf16 with fp32 '
                           'master weights, gradients and optimizer state (INTEGRATION.md, "precision")')
        self.set_optimizers()
        if getattr(params, 'multi_gpu', False):
            logger.info('Using m3p_amd.distributed.DataParallel (bucketed RCCL all-reduce) ...')
            for name in self.MODEL_NAMES:
                wrapped = DataParallel(getattr(self, name))
                setattr(self, name, wrapped)
                for opt in self.optimizers.values():
                    opt.grad_scale = 1.0 / wrapped.world
        # probability of masking out / keeping / randomising the words to predict
        params.pred_probs = torch.FloatTensor([getattr(params, 'word_mask', 0.8), getattr(params, 'word_keep', 0.1),
                                               getattr(params, 'word_rand', 0.1)])
        self.metrics = []
        for m in getattr(params, 'validation_metrics', '').split(','):
            if m != '':
                self.metrics.append((m[1:], False) if m[0] == '_' else (m, True))
        self.best_metrics = {metric: (-1e12 if biggest else 1e12) for metric, biggest in self.metrics}
        self.epoch = 0
        self.n_iter = 0
        self.n_total_iter = 0
        self.n_sentences = 0
        self.stats = OrderedDict([('processed_s', 0), ('processed_w', 0)] + [(k, []) for k in _stat_names(params)])
        self.last_time = time.time()
        self._pending_w = []
        self.reload_checkpoint()
        parse_lambda_config(params)

    def set_parameters(self):
        """xtrainer.py:168-184."""
        named = []
        for name in self.MODEL_NAMES:
            named.extend([(k, p) for k, p in getattr(self, name).named_parameters() if p.requires_grad])
        self.parameters = {'model': [p for k, p in named]}
        assert len(self.parameters['model']) >= 1

    def set_optimizers(self):
        """xtrainer.py:186-203."""
        self.optimizers = {'model': get_optimizer(self.parameters['model'], self.params.optimizer)}

    def _stat(self, key, value):
        self.stats.setdefault(key, []).append(value.detach() if torch.is_tensor(value) else value)

    def _dp_plan(self, vocab_dense, expect=()):
        """Data parallelism: tell the reducer whether this step type has an MLM head (a dense gradient for the
        vocabulary matrix), and mark the heads this step type MAY train as touched on every rank - whether a head
        actually runs depends on the rank's batch (any masked word / region?), and ranks must agree on the ranges
        the optimizer steps and zeroes."""
        model = getattr(self, 'model')
        if isinstance(model, DataParallel) and not model.single:
            model.plan_step(vocab_dense)
            arena = model.module.arena()
            for head in expect:
                arena.plan(*_REGION_HEADS[head])       # (announced, not written: the MLM head's first product may still STORE)

    def optimize(self, loss):
        """xtrainer.py:205-243 without the host round trips: backward -> (bucketed
        all-reduce overlapped with it) -> global-norm clip -> Adam -> zero_grad, the last
        three as one fused pass over the arenas."""
        params = self.params
        accumulate = max(int(getattr(params, 'accumulate_gradients', 1)), 1)
        boundary = self.n_iter % accumulate == 0   # xtrainer.py:231
        model = getattr(self, 'model')
        if accumulate > 1 and not boundary and isinstance(model, DataParallel):
            with model.no_sync():
                loss.backward()
            return
        loss.backward()
        if not boundary:
            return
        for opt in self.optimizers.values():
            if params.clip_grad_norm > 0:
                opt.clip_grad_norm(params.clip_grad_norm)
            opt.step()

    def iter(self):
        """xtrainer.py:245-252."""
        self.n_iter += 1
        self.n_total_iter += 1
        update_lambdas(self.params, self.n_total_iter)
        self.print_stats()

    def print_stats(self):
        """Every 5 iterations: mean of each loss since the last print, learning rates, and the
        throughput meter sent/s = processed_s / elapsed, words/s likewise (xtrainer.py:254-289 -
        the reference's definition of the metric bench.py reports).  The only place the step
        statistics are read back from the device."""
        if self.n_iter % 5 != 0:
            return
        if self._pending_w:
            self.stats['processed_w'] += int(torch.stack([w.reshape(()) for w in self._pending_w]).sum().item())
            self._pending_w = []
        means = []
        for key, vals in self.stats.items():
            if isinstance(vals, list) and vals:
                host = torch.stack([v.float().reshape(()) for v in vals]).tolist() if torch.is_tensor(vals[0]) else vals
                means.append('%s: %7.4f' % (key, float(np.mean(host))))
                del vals[:]
        rates = ' - ' + ''.join(' - %s LR: %s' % (name, ' / '.join('%.4e' % g['lr'] for g in opt.param_groups))
                                for name, opt in self.optimizers.items())
        now = time.time()
        elapsed = now - self.last_time
        speed = '%7.2f sent/s - %8.2f words/s - ' % (self.stats['processed_s'] / elapsed, self.stats['processed_w'] / elapsed)
        self.stats['processed_s'] = self.stats['processed_w'] = 0
        self.last_time = now
        logger.info('%7i - ' % self.n_iter + speed + ' || '.join(means) + rates)

    # ------------------------------------------------------------------ text batches (xtrainer.py:436-509)
    def get_cross_lingual_iterator(self, iter_name, lang1, lang2, stream):
        """Data-layer contract (the datasets themselves are the reference's): ``data['mono_stream'][lang]['train']``
        / ``data['mono'][lang]['train']`` (``data['text']`` under ``is_ntg``) / ``data['para'][(l1, l2)]['train']`` expose ``get_iterator(...)``
        yielding ``(x, lengths)`` (or a pair of those for parallel data)."""
        logger.info('Creating new training data iterator (%s) ...' % ','.join(
            str(v) for v in (iter_name, lang1, lang2) if v is not None))
        if lang2 is None:
            if stream:
                iterator = self.data['mono_stream'][lang1]['train'].get_iterator(shuffle=True)
            else:                                   # text-to-text generation reads its own table (:446-450)
                table = 'text' if getattr(self.params, 'is_ntg', False) else 'mono'
                iterator = self.data[table][lang1]['train'].get_iterator(
                    shuffle=True, group_by_size=self.params.group_by_size, n_sentences=-1)
        else:
            pair = (lang1, lang2) if lang1 < lang2 else (lang2, lang1)
            iterator = self.data['para'][pair]['train'].get_iterator(
                shuffle=True, group_by_size=self.params.group_by_size, n_sentences=-1)
        self.iterators[(iter_name, lang1, lang2)] = iterator
        return iterator

    def get_cross_lingual_batch(self, iter_name, lang1, lang2=None, stream=False):
        assert lang1 in self.params.langs and (lang2 is None or lang2 in self.params.langs)
        iterator = self.iterators.get((iter_name, lang1, lang2))
        if iterator is None:
            iterator = self.get_cross_lingual_iterator(iter_name, lang1, lang2, stream)
        try:
            x = next(iterator)
        except StopIteration:
            x = next(self.get_cross_lingual_iterator(iter_name, lang1, lang2, stream))
        return x if lang2 is None or lang1 < lang2 else x[::-1]

    def generate_batch(self, lang1, lang2, name):
        """xtrainer.py:485-509: the monolingual stream (MLM), a sentence next to its noised copy (lang1 == lang2), or a
        parallel pair joined with reset positions and per-token language ids (TLM)."""
        params = self.params
        lang1_id = params.lang2id[lang1]
        if lang2 is None:
            x, lengths = self.get_cross_lingual_batch(name, lang1, stream=True)
            langs = x.clone().fill_(lang1_id) if params.n_langs > 1 else None
            return x, lengths, None, langs, (None, None)
        lang2_id = params.lang2id[lang2]
        if lang1 == lang2:
            x2, len2 = self.get_cross_lingual_batch(name, lang1)
            x1, len1 = self.add_noise(x2, len2)
        else:
            (x1, len1), (x2, len2) = self.get_cross_lingual_batch(name, lang1, lang2)
        x, lengths, positions, langs = concat_batches(x1, len1, lang1_id, x2, len2, lang2_id, params.pad_index,
                                                      params.eos_index, reset_positions=lang1 != lang2)
        return x, lengths, positions, langs, (len1, len2)

    def round_batch(self, x, lengths, positions, langs):
        """xtrainer.py:654-692."""
        return masking.round_batch(x, lengths, positions, langs, self.params)

    def mask_out(self, x, lengths):
        """xtrainer.py:385-434."""
        return masking.mask_out(x, lengths, self.params)

    def mlm_step(self, lang1, lang2, lambda_coeff):
        """Masked word prediction on a text batch (xtrainer.py:734-770)."""
        assert lambda_coeff >= 0
        if lambda_coeff == 0:
            return
        x, lengths, positions, langs, _ = self.generate_batch(lang1, lang2, 'pred')
        x, lengths, positions, langs, _ = self.round_batch(x, lengths, positions, langs)
        x, y, pred_mask = self.mask_out(x, lengths)
        return self.mlm_step_on_batch(x, lengths, pred_mask, y, lang1, lambda_coeff, langs=langs, positions=positions,
                                      stat=None if lang2 is None else 'MLM-%s-%s' % (lang1, lang2))

    def mlm_step_on_batch(self, x, lengths, pred_mask, y, lang='en', lambda_coeff=1, langs=None, positions=None, stat=None):
        """Loss path of mlm_step on an already masked batch (:751-770)."""
        model = self.model
        model.train()
        self._dp_plan(True, expect=('mlm',))
        n_words = pred_mask.sum()
        x, y, pred_mask, lengths = to_cuda(x, y, pred_mask, lengths)
        # (langs is None unless params.n_langs > 1: then the stream adds the language embeddings, transformer.py:1059-1060;
        #  positions only for TLM batches, whose second sentence restarts at 0)
        tensor = model('crossfwd', stream_='text', x=x, lengths=lengths, positions=positions, langs=langs, causal=False)
        _, loss = model('predict', tensor=tensor, pred_mask=pred_mask, y=y, get_scores=False)
        self._stat(stat or 'MLM-%s' % lang, loss)
        self.optimize(lambda_coeff * loss)
        self.n_sentences += self.params.batch_size
        self.stats['processed_s'] += lengths.size(0)
        self._pending_w.append(n_words)
        return loss

    # ------------------------------------------------------------------ checkpoints (xtrainer.py:511-650)

    def add_noise(self, words, lengths):
        """xtrainer.py:376-383 (word_shuffle + word_dropout under np.random, bit-identical: m3p_amd/masking.py)."""
        return masking.add_noise(words, lengths, self.params)

    def mt_step(self, lang1, lang2, lambda_coeff):
        """Machine translation step, or - lang1 == lang2 - denoising auto-encoding of a monolingual batch
        (xtrainer.py:1383-1441)."""
        assert lambda_coeff >= 0
        if lambda_coeff == 0:
            return
        if lang1 == lang2:
            (x1, len1) = self.get_cross_lingual_batch('ae', lang1)
            (x2, len2) = (x1, len1)
            (x1, len1) = self.add_noise(x1, len1)
        else:
            (x1, len1), (x2, len2) = self.get_cross_lingual_batch('mt', lang1, lang2)
        return self.mt_step_on_batch(x1, len1, x2, len2, lang1, lang2, lambda_coeff)

    def mt_step_on_batch(self, x1, len1, x2, len2, lang1, lang2, lambda_coeff=1, stat=None):
        """Loss path of mt_step on a parallel batch (:1410-1441): the model encodes the source sentence (non-causal text
        stream with its language embedding) and decodes the target with teacher forcing (causal stream + attention over
        the encoding); word t + 1 is predicted from position t."""
        params = self.params
        model = self.model
        model.train()
        self._dp_plan(True, expect=('mlm',))
        langs1 = x1.clone().fill_(params.lang2id[lang1])
        langs2 = x2.clone().fill_(params.lang2id[lang2])
        alen = torch.arange(int(len2.max()), dtype=torch.long, device=len2.device)
        pred_mask = alen[:, None] < len2[None] - 1           # nothing is predicted from a sentence's last word
        y = x2[1:].masked_select(pred_mask[:-1])
        n_words = int((len2 - 1).sum())
        assert len(y) == n_words
        x1, len1, langs1, x2, len2, langs2, y, pred_mask = to_cuda(x1, len1, langs1, x2, len2, langs2, y, pred_mask)
        enc1 = model('crossfwd', stream_='text', x=x1, lengths=len1, langs=langs1, causal=False)
        enc1 = enc1.transpose(0, 1)
        dec2 = model('crossfwd', stream_='text', x=x2, lengths=len2, langs=langs2, causal=True, src_enc=enc1, src_len=len1)
        _, loss = model('predict', tensor=dec2, pred_mask=pred_mask, y=y, get_scores=False)
        self._stat(stat or (('AE-%s' % lang1) if lang1 == lang2 else ('MT-%s-%s' % (lang1, lang2))), loss)
        self.optimize(lambda_coeff * loss)
        self.n_sentences += params.batch_size
        self.stats['processed_s'] += len2.size(0)
        self.stats['processed_w'] += n_words
        return loss.detach()

    def bart_token_mask_sent(self, x, lengths, min_len=100000):
        """xtrainer.py:1318-1381."""
        return masking.bart_token_mask_sent(x, lengths, self.params, min_len)

    def restricted_mask_sent(self, x, lengths, min_len=100000):
        """xtrainer.py:1269-1316 (the MASS batch)."""
        return masking.restricted_mask_sent(x, lengths, self.params, min_len)

    def bart_mlm_step(self, lang1, lang2, lambda_coeff):
        """Text-infilling denoising step (xtrainer.py:1595-1646): one span of every sentence of a monolingual stream batch
        collapses into a <mask> (``bart_token_mask_sent``, optionally followed by ``add_noise``), the model encodes that and
        decodes the whole original sentence with teacher forcing.  Both sides carry lang1's id (:1608-1609)."""
        assert lambda_coeff >= 0
        if lambda_coeff == 0:
            return
        params = self.params
        model = self.model
        model.train()
        self._dp_plan(True, expect=('mlm',))
        x, lengths, positions, langs, _ = self.generate_batch(lang1, lang2, 'pred')
        x, lengths, positions, langs, _ = self.round_batch(x, lengths, positions, langs)
        x1, len1, x2, len2, y, pred_mask, _pos = self.bart_token_mask_sent(x, lengths)
        if getattr(params, 'use_noise', False):
            x1, len1 = self.add_noise(x1, len1)
        langs1 = x1.clone().fill_(params.lang2id[lang1])
        langs2 = x2.clone().fill_(params.lang2id[lang1])
        n_words = int(pred_mask.sum())
        x1, x2, len1, len2, y, pred_mask, langs1, langs2 = to_cuda(x1, x2, len1, len2, y, pred_mask, langs1, langs2)
        enc1 = model('crossfwd', stream_='text', x=x1, lengths=len1, langs=langs1, causal=False).transpose(0, 1)
        dec2 = model('crossfwd', stream_='text', x=x2, lengths=len2, langs=langs2, causal=True, src_enc=enc1, src_len=len1)
        _, loss = model('predict', tensor=dec2, pred_mask=pred_mask, y=y, get_scores=False)
        self._stat(('M-BART-%s' % lang1) if lang2 is None else ('MLM-%s-%s' % (lang1, lang2)), loss)
        self.optimize(lambda_coeff * loss)
        self.n_sentences += params.batch_size
        self.stats['processed_s'] += lengths.size(0)
        self.stats['processed_w'] += n_words
        return loss.detach()

    def bart_mass_step(self, lang1, lang2, lambda_coeff):
        """MASS step (xtrainer.py:1648-1697): a span of every sentence is hidden in the encoder input (80 / 10 / 10 rule,
        ``restricted_mask_sent``); the decoder reads the words before the hidden ones AT THEIR ORIGINAL POSITIONS, may not
        look at the source's <mask> positions (``enc_mask``) and predicts the hidden words."""
        assert lambda_coeff >= 0
        if lambda_coeff == 0:
            return
        params = self.params
        model = self.model
        model.train()
        self._dp_plan(True, expect=('mlm',))
        x, lengths, positions, langs, _ = self.generate_batch(lang1, lang2, 'pred')
        x, lengths, positions, langs, _ = self.round_batch(x, lengths, positions, langs)
        x1, len1, x2, len2, y, pred_mask, positions = self.restricted_mask_sent(x, lengths)
        return self.mass_step_on_batch(x1, len1, x2, len2, y, pred_mask, positions, lang1, lang2, lambda_coeff, n_sent=lengths.size(0))

    def mass_step_on_batch(self, x1, len1, x2, len2, y, pred_mask, positions, lang1, lang2=None, lambda_coeff=1, n_sent=None):
        """Loss path of bart_mass_step (:1668-1697) on a batch of ``restricted_mask_sent``."""
        params = self.params
        model = self.model
        model.train()
        langs1 = x1.clone().fill_(params.lang2id[lang1])
        langs2 = x2.clone().fill_(params.lang2id[lang1])
        n_words = int(pred_mask.sum())
        enc_mask = x1.ne(params.mask_index).transpose(0, 1)
        x1, x2, len1, len2, y, pred_mask, positions, langs1, langs2, enc_mask = to_cuda(
            x1, x2, len1, len2, y, pred_mask, positions, langs1, langs2, enc_mask)
        enc1 = model('crossfwd', stream_='text', x=x1, lengths=len1, langs=langs1, causal=False).transpose(0, 1)
        dec2 = model('crossfwd', stream_='text', x=x2, lengths=len2, langs=langs2, causal=True, src_enc=enc1, src_len=len1,
                     positions=positions, enc_mask=enc_mask)
        _, loss = model('predict', tensor=dec2, pred_mask=pred_mask, y=y, get_scores=False)
        self._stat(('M-MASS-%s' % lang1) if lang2 is None else ('M-MASS-%s-%s' % (lang1, lang2)), loss)
        self.optimize(lambda_coeff * loss)
        self.n_sentences += params.batch_size
        self.stats['processed_s'] += len1.size(0) if n_sent is None else n_sent
        self.stats['processed_w'] += n_words
        return loss.detach()

    def ntg_step(self, lang1='en', lang2=None, lambda_coeff=1):
        """Text-to-text generation step (xtrainer.py:2596-2645; train_x.py:443-445 over ``text_steps``): a
        (source, target) batch of one language from ``ntg_collate`` - the translation step's loss path with the same language
        id on both sides and no input noise."""
        assert lambda_coeff >= 0
        if lambda_coeff == 0:
            return
        (x1, len1), (x2, len2) = self.get_cross_lingual_batch('ntg', lang1, None)
        return self.mt_step_on_batch(x1, len1, x2, len2, lang1, lang1, lambda_coeff, stat='NTG-%s' % lang1)

    def ic_step(self, dataset='coco', input_stream='img', lambda_coeff=1):
        """Captioning step (xtrainer.py:1443-1515) on a ``('txt2img', dataset, 'img')`` batch:
        ``(x2, len2), (x1, x1_mask, img_loc, img_id)``."""
        assert lambda_coeff >= 0
        if lambda_coeff == 0:
            return
        batch = self.get_batch('txt2img', dataset, input_stream)
        (x2, len2), (x1, x1_mask, img_loc, _img_id) = batch
        return self.ic_step_on_batch(x2, len2, x1, x1_mask, img_loc, dataset, input_stream, lambda_coeff)

    def ic_step_on_batch(self, x2, len2, x1, x1_mask, img_loc, dataset='coco', input_stream='img', lambda_coeff=1, stat=None):
        """Loss path of ic_step (:1466-1515): the model encodes the regions (image-only stream) and decodes the caption with
        teacher forcing over that encoding.  x1 (B, R, 2048), x1_mask (B, R), img_loc (B, R, 5) as the collate emits them."""
        params = self.params
        model = self.model
        model.train()
        self._dp_plan(True, expect=('mlm',))
        ft = getattr(params, 'ft_lgs', None) or []
        lang_id = params.lang2id[ft[0]] if len(ft) > 0 else params.lang2id['en']
        langs = x2.clone().fill_(lang_id)
        alen = torch.arange(int(len2.max()), dtype=torch.long, device=len2.device)
        pred_mask = alen[:, None] < len2[None] - 1
        y = x2[1:].masked_select(pred_mask[:-1])
        n_words = int((len2 - 1).sum())
        assert len(y) == n_words
        len1 = x1_mask.sum(dim=1)
        x1 = x1.transpose(0, 1)
        img_loc = img_loc.transpose(0, 1)
        langs_img = x1_mask.transpose(0, 1).clone().long().fill_(lang_id)
        x1, len1, img_loc, x2, len2, y, langs, langs_img, pred_mask = to_cuda(x1, len1, img_loc, x2, len2, y, langs, langs_img, pred_mask)
        enc1 = model('crossfwd', stream_='img', x=x1, lengths=len1, langs=langs_img, causal=False, image_loc=img_loc,
                     refine_image=getattr(params, 'refine_image', False))
        enc1 = enc1.transpose(0, 1)
        dec2 = model('crossfwd', stream_='text', x=x2, lengths=len2, langs=langs, causal=True, src_enc=enc1, src_len=len1)
        _, loss = model('predict', tensor=dec2, pred_mask=pred_mask, y=y, get_scores=False)
        self._stat(stat or 'IC-%s-%s' % (dataset, input_stream), loss)
        self.optimize(lambda_coeff * loss)
        self.n_sentences += params.batch_size
        self.stats['processed_s'] += len2.size(0)
        self.stats['processed_w'] += n_words
        return loss.detach()
    def bart_img_noise(self, object_features, loc_features, img_mask):
        """xtrainer.py:1734-1744."""
        return masking.bart_img_noise(object_features, loc_features, img_mask)

    def bart_img_step(self, dataset='coco', input_stream='img', token_mask=False, lambda_coeff=1):
        """Image denoising step (xtrainer.py:1746-1808; train_x.py:462-463 over ``cross_ae_steps``): the captioning pass on a
        batch whose region features went through ``bart_img_noise`` (spans of regions blanked and collapsed, fewer regions)."""
        assert lambda_coeff >= 0
        if lambda_coeff == 0:
            return
        (x2, len2), (x_img, x_img_mask, img_loc, _img_id) = self.get_batch('ida', dataset, input_stream)
        x_img, img_loc, x_img_mask = self.bart_img_noise(x_img, img_loc, x_img_mask)
        return self.ic_step_on_batch(x2, len2, x_img, x_img_mask, img_loc, dataset, input_stream, lambda_coeff,
                                     stat='IDA-%s' % dataset)

    def mt_ic_step(self, dataset='coco', input_stream='img', lambda_coeff=1):
        """Multimodal translation step (xtrainer.py:1517-1593) on a ``mt_caption_collate`` batch
        ``(x_src, len_src), (x2, len2), (x1, x1_mask, img_loc, img_id)``."""
        assert lambda_coeff >= 0
        if lambda_coeff == 0:
            return
        (x_src, len_src), (x2, len2), (x1, x1_mask, img_loc, _img_id) = self.get_batch('txt2img', dataset, input_stream)
        return self.mt_ic_step_on_batch(x_src, len_src, x2, len2, x1, x1_mask, img_loc, dataset, input_stream, lambda_coeff)

    def mt_ic_step_on_batch(self, x_src, len_src, x2, len2, x1, x1_mask, img_loc, dataset='coco', input_stream='img',
                            lambda_coeff=1):
        """Loss path of mt_ic_step (:1540-1593): the source sentence is encoded together with the image regions (``jointfwd``,
        language ids ignored there) - or alone with its language embedding when ``params.mt_only_text`` - and the target is
        decoded with teacher forcing over that encoding (``src_len`` = source words + regions).  Languages come from
        ``params.ft_lgs`` = [source, target]."""
        params = self.params
        model = self.model
        model.train()
        self._dp_plan(True, expect=('mlm',))
        lang_src = x_src.clone().fill_(params.lang2id[params.ft_lgs[0]])
        langs = x2.clone().fill_(params.lang2id[params.ft_lgs[1]])
        alen = torch.arange(int(len2.max()), dtype=torch.long, device=len2.device)
        pred_mask = alen[:, None] < len2[None] - 1
        y = x2[1:].masked_select(pred_mask[:-1])
        n_words = int((len2 - 1).sum())
        assert len(y) == n_words
        len1 = x1_mask.sum(dim=1)
        x1 = x1.transpose(0, 1)
        img_loc = img_loc.transpose(0, 1)
        x1, len1, img_loc, x2, len2, y, langs, lang_src, x_src, len_src, pred_mask = to_cuda(
            x1, len1, img_loc, x2, len2, y, langs, lang_src, x_src, len_src, pred_mask)
        if getattr(params, 'mt_only_text', False):
            enc1 = model('crossfwd', stream_='text', x=x_src, lengths=len_src, langs=lang_src, causal=False)
            len_all = len_src
        else:
            enc1 = model('jointfwd', x=x_src, lengths=len_src, x_img=x1, lengths_img=len1, causal=False, langs=None,
                         image_loc=img_loc, refine_image=getattr(params, 'refine_image', False))
            len_all = len_src + len1
        enc1 = enc1.transpose(0, 1)
        dec2 = model('crossfwd', stream_='text', x=x2, lengths=len2, langs=langs, causal=True, src_enc=enc1, src_len=len_all)
        _, loss = model('predict', tensor=dec2, pred_mask=pred_mask, y=y, get_scores=False)
        self._stat('IC-%s-%s' % (dataset, input_stream), loss)
        self.optimize(lambda_coeff * loss)
        self.n_sentences += params.batch_size
        self.stats['processed_s'] += len2.size(0)
        self.stats['processed_w'] += n_words
        return loss.detach()

    def _sync_master(self):
        """Sharded data parallelism keeps the fp32 master of the big matrices on its owner rank between steps: gather it
        (a collective - EVERY rank comes through here, before any master-rank test) so that whoever saves holds all of it."""
        for n in self.MODEL_NAMES:
            hook = getattr(_unwrap(getattr(self, n)), 'ddp_hook', None)
            if hook is not None:
                hook.materialize_master()

    def _state_dicts(self):
        # sharded data parallelism (distributed.py, zero1): since round 4 the forward-side gathers move the bf16 working copy
        # only, and the fp32 master of the big matrices stays on its owner rank until DataParallel.materialize_master() - a
        # COLLECTIVE every rank must run (save_model / save_checkpoint / save_periodic / save_best_model / end_epoch do, through
        # _materialize(), before their master-rank test).  A direct call of this method or of model.state_dict() without it
        # raises under zero1 instead of handing out stale weights.  params_ready(None) below is the
        # stream-level wait for the gathers that ARE in flight.  The Adam MOMENTS stay sharded: a checkpoint holds this rank's
        # shards of them (zeros elsewhere); neither the reference's reload (xtrainer.py:586-592) nor ours reads them back.
        for n in self.MODEL_NAMES:
            hook = getattr(_unwrap(getattr(self, n)), 'ddp_hook', None)
            if hook is not None:
                hook.params_ready(None)
        return {n: {k: v.detach().cpu().clone() for k, v in _unwrap(getattr(self, n)).state_dict().items()}
                for n in self.MODEL_NAMES}

    def _params_dict(self):
        return {k: v for k, v in self.params.__dict__.items()}

    def save_model(self, name):
        """{'model': state_dict, 'params': dict} (:511-529), written by the master rank only."""
        self._sync_master()
        if not getattr(self.params, 'is_master', True):
            return None
        path = os.path.join(self.params.dump_path, '%s.pth' % name)
        logger.info('Saving models to %s ...' % path)
        data = self._state_dicts()
        data['params'] = self._params_dict()
        torch.save(data, path)
        return path

    def save_checkpoint(self, name='checkpoint', include_optimizers=True):
        """:531-560: model + optimizer ``state_dict()`` (param_groups with num_updates / lr, and the Adam moments)
        + epoch counters + best metrics; master rank only."""
        self._sync_master()
        if not getattr(self.params, 'is_master', True):
            return None
        path = os.path.join(self.params.dump_path, '%s.pth' % name)
        logger.info('Saving %s to %s ...' % (name, path))
        data = {'epoch': self.epoch, 'n_total_iter': self.n_total_iter, 'best_metrics': self.best_metrics,
                'best_stopping_criterion': self.best_stopping_criterion}
        data.update(self._state_dicts())
        if include_optimizers:
            for n, opt in self.optimizers.items():
                sd = opt.state_dict()
                for st in sd['state'].values():      # moments are views of the flat arenas: detach them from it
                    for k, v in st.items():
                        if torch.is_tensor(v):
                            st[k] = v.detach().cpu().clone()
                data['%s_optimizer' % n] = sd
        data['params'] = self._params_dict()
        torch.save(data, path)
        return path

    def reload_checkpoint(self, path=None):
        """:562-599: looks for dump_path/checkpoint.pth, else params.reload_checkpoint; restores the weights
        (``module.`` prefixes of a DDP-saved file stripped), and - like the reference - only ``num_updates`` / lr of
        the optimizer, then the epoch counters and best metrics.  ``path`` overrides the search (extension)."""
        params = self.params
        if path is None:
            path = os.path.join(getattr(params, 'dump_path', ''), 'checkpoint.pth')
            if not os.path.isfile(path):
                path = getattr(params, 'reload_checkpoint', '')
                if path == '':
                    return
                assert os.path.isfile(path), path
        logger.warning('Reloading checkpoint from %s ...' % path)
        data = torch.load(path, map_location='cpu', weights_only=False)
        for n in self.MODEL_NAMES:
            sd = data[n]
            if all(k.startswith('module.') for k in sd):
                sd = {k[len('module.'):]: v for k, v in sd.items()}
            _unwrap(getattr(self, n)).load_state_dict(sd)
        for n, opt in self.optimizers.items():
            saved = data.get('%s_optimizer' % n)
            if saved is None:
                continue
            for gid, g in enumerate(opt.param_groups):
                if 'num_updates' in g and 'num_updates' in saved['param_groups'][gid]:
                    g['num_updates'] = saved['param_groups'][gid]['num_updates']
                    g['lr'] = opt.get_lr_for_step(g['num_updates'])
        self.epoch = data['epoch'] + 1
        self.n_total_iter = data['n_total_iter']
        self.best_metrics = data['best_metrics']
        self.best_stopping_criterion = data['best_stopping_criterion']
        logger.warning('Checkpoint reloaded. Resuming at epoch %i / iteration %i ...' % (self.epoch, self.n_total_iter))

    def save_periodic(self):
        """:601-608."""
        self._sync_master()
        every = getattr(self.params, 'save_periodic', 0)
        if getattr(self.params, 'is_master', True) and every > 0 and self.epoch % every == 0:
            self.save_model('periodic-%i' % self.epoch)

    def save_best_model(self, scores):
        """:610-625: one 'best-<metric>' model + checkpoint per validation metric that improved."""
        self._sync_master()
        if not getattr(self.params, 'is_master', True):
            return
        for metric, biggest in self.metrics:
            if metric not in scores:
                logger.warning('Metric "%s" not found in scores!' % metric)
                continue
            sign = 1 if biggest else -1
            if sign * scores[metric] > sign * self.best_metrics[metric]:
                self.best_metrics[metric] = scores[metric]
                logger.info('New best score for %s: %.6f' % (metric, scores[metric]))
                self.save_model('best-%s' % metric)
                self.save_checkpoint('best-%s' % metric, include_optimizers=True)

    def end_epoch(self, scores):
        """:627-650: early stopping on the stopping criterion, then the rolling checkpoint."""
        self._sync_master()
        if self.stopping_criterion is not None and (getattr(self.params, 'is_master', True)
                                                    or not self.stopping_criterion[0].endswith('_mt_bleu')):
            metric, biggest = self.stopping_criterion
            assert metric in scores, metric
            sign = 1 if biggest else -1
            if sign * scores[metric] > sign * self.best_stopping_criterion:
                self.best_stopping_criterion = scores[metric]
                logger.info('New best validation score: %f' % self.best_stopping_criterion)
                self.decrease_counts = 0
            else:
                logger.info('Not a better validation score (%i / %i).' % (self.decrease_counts, self.decrease_counts_max))
                self.decrease_counts += 1
            if self.decrease_counts > self.decrease_counts_max:
                logger.info('Stopping criterion has been below its best value for more than %i epochs. '
                            'Ending the experiment...' % self.decrease_counts_max)
                if getattr(self.params, 'multi_gpu', False) and 'SLURM_JOB_ID' in os.environ:
                    os.system('scancel ' + os.environ['SLURM_JOB_ID'])
                raise SystemExit(0)
        self.save_checkpoint('checkpoint', include_optimizers=True)
        self.epoch += 1


class XTrainer(Trainer):
    def __init__(self, model, data, params):
        """xtrainer.py:1130-1146."""
        self.MODEL_NAMES = ['model']
        self.model = model
        self.data = data
        self.params = params
        super().__init__(data, params)

    # ------------------------------------------------------------------ cross-modal batches (xtrainer.py:1148-1206)
    def get_iterator(self, iter_name, lang1, lang2):
        """A DataLoader over ``data['cross_modal'][(dataset, 'img')]['train']`` with the pre-training or the
        retrieval / captioning / multimodal-translation / sliding-window collate the run's flags select; distributed
        runs shard it with a DistributedSampler."""
        from torch.utils.data import DataLoader, RandomSampler
        from torch.utils.data.distributed import DistributedSampler
        params = self.params
        logger.info('Creating new training data iterator (%s) ...' % ','.join(
            str(v) for v in (iter_name, lang1, lang2) if v is not None))
        dataset = self.data['cross_modal'][(lang1, lang2)]['train']
        if lang1 in ('google', 'sbu') and hasattr(dataset, 'update'):
            dataset.update(self.epoch)                   # conceptual-captions shards rotate per epoch
        if lang1 == 'flicker' and hasattr(dataset, 'update_captions'):
            dataset.update_captions()
        sampler = RandomSampler(dataset) if getattr(params, 'n_gpu_per_node', 1) == 1 else DistributedSampler(dataset)
        if getattr(params, 'is_generation', False):            # xtrainer.py:1164-1181
            collate = mt_caption_collate if getattr(params, 'is_mt', False) else caption_collate
        elif getattr(params, 'is_pretrain', False):
            collate = retrieval_pretrain_collate
        else:
            collate = slide_collate if getattr(params, 'is_slide', False) else retrieval_collate
        loader = DataLoader(dataset, batch_size=params.batch_size, sampler=sampler, collate_fn=collate,
                            num_workers=getattr(params, 'num_workers', 0))
        for batch in loader:
            yield batch

    def get_batch(self, iter_name, lang1, lang2=None):
        assert lang2 == 'img'
        key = (iter_name, lang1, lang2)
        iterator = self.iterators.get(key)
        if iterator is None:
            iterator = self.iterators[key] = self.get_iterator(iter_name, lang1, lang2)
        try:
            return next(iterator)
        except StopIteration:
            if getattr(self.params, 'is_pretrain', False):
                self.iterators = {}
            iterator = self.iterators[key] = self.get_iterator(iter_name, lang1, lang2)
            return next(iterator)

    # ------------------------------------------------------------------ masks / losses
    def get_mask_(self, x, _labels):
        """xtrainer.py:2226-2232: mask = labels != -1; targets = labels[labels > 0]."""
        pred_mask = (_labels != -1)
        y = _labels[_labels > 0]
        return y, pred_mask

    def _itm_loss(self, relation_scores, pos_labels):
        """xtrainer.py:2357-2372 kept on the device: CE over groups of sample_n + BCE vs one-hot."""
        params = self.params
        dev = relation_scores.device
        pos = to_cuda(torch.as_tensor(np.asarray(pos_labels), dtype=torch.long).reshape(-1))[0] if dev.type == 'cuda' else \
            torch.as_tensor(np.asarray(pos_labels), dtype=torch.long).reshape(-1)
        if dev.type == 'cuda':      # one fused launch (loss + its gradient) instead of ~20 elementwise ones
            from . import functional as Fn
            return Fn.ItmLossFn.apply(relation_scores, pos.contiguous(), params.sample_n,
                                      float(params.multi_cls_loss_weight), float(params.bin_cls_loss_weight))
        onehot = F.one_hot(pos, params.sample_n).float().view(-1)
        scores = relation_scores.float()
        loss = 0
        if params.multi_cls_loss_weight != 0:
            loss = loss + params.multi_cls_loss_weight * F.cross_entropy(scores.view(-1, params.sample_n), pos)
        if params.bin_cls_loss_weight != 0:
            loss = loss + params.bin_cls_loss_weight * F.binary_cross_entropy_with_logits(scores.view(-1), onehot)
        return loss

    # ------------------------------------------------------------------ hot steps
    def pretrain_rel_step(self, dataset='coco', input_stream='img'):
        """xtrainer.py:1879-1886."""
        p = self.params
        t2i_batch, i2t_batch = self.get_batch('rel', dataset, input_stream)
        if p.t2i_flag:
            self.pretrain_under_step(t2i_batch, dataset, 't2i', 'en', p.lambda_t2i, p.lambda_mlm, p.lambda_mrm, p.lambda_mrfr)
        if p.i2t_flag:
            self.pretrain_under_step(i2t_batch, dataset, 'i2t', 'en', p.lambda_i2t, p.lambda_mlm, p.lambda_mrm, p.lambda_mrfr)

    def rel_step(self, dataset='coco', input_stream='img', lambda_1=1, lambda_2=1):
        """xtrainer.py:1867-1877; with ``params.is_freelb`` every task's step is preceded by its adversarial one - the
        reference hands BOTH of those the t2i batch and lambda_1 (:1871, :1875), kept."""
        t2i_batch, i2t_batch = self.get_batch('rel', dataset, input_stream)
        freelb = getattr(self.params, 'is_freelb', False)
        if self.params.t2i_flag:
            if freelb:
                self.freelb_t2i_step(t2i_batch, dataset, lambda_1)
            self.t2i_step(t2i_batch, dataset, lambda_1)
        if self.params.i2t_flag:
            if freelb:
                self.freelb_i2t_step(t2i_batch, dataset, lambda_1)
            self.i2t_step(i2t_batch, dataset, lambda_2)

    # ---- FreeLB adversarial fine-tuning of the matching steps (xtrainer.py:2021-2224, :2700-2851)
    @staticmethod
    def _uniform_like(t):
        """U(-1, 1) noise of t's shape, drawn on the host from torch's CPU generator (the reference draws on whatever device
        the embeddings live on; drawn here where a seeded CPU run of the reference draws it, so that the two can be compared)."""
        return torch.zeros(t.shape, dtype=torch.float32).uniform_(-1, 1).to(t.device)

    def deal_freelb_delta(self, model, input_ids, input_lengths, adv_init_mag=1e-4, norm_type='l2'):
        """:2700-2722: the word embeddings of the batch and a random perturbation of magnitude adv_init_mag / sqrt(len * d)."""
        embeds_init = _unwrap(model).embeddings(input_ids)
        if adv_init_mag > 0:
            if norm_type == 'l2':
                dims = input_lengths * embeds_init.size(-1)
                mag = adv_init_mag / torch.sqrt(dims.float())
                delta = (self._uniform_like(embeds_init) * mag.view(-1, 1, 1)).detach()
            else:
                assert norm_type == 'linf'
                delta = self._uniform_like(embeds_init) * adv_init_mag
        else:
            delta = torch.zeros_like(embeds_init, dtype=torch.float32)
        return embeds_init, delta

    def deal_image_freelb_delta(self, image_feature, adv_init_mag=1e-4, norm_type='l2'):
        """:2724-2735 (the magnitude is per FIRST index of the (R, B, 2048) features, as the reference has it)."""
        if adv_init_mag > 0:
            if norm_type == 'l2':
                mag = adv_init_mag / math.sqrt(image_feature.size(-1))
                return (self._uniform_like(image_feature) * mag).detach()
            assert norm_type == 'linf'
            return self._uniform_like(image_feature) * adv_init_mag
        return torch.zeros_like(image_feature, dtype=torch.float32)

    @staticmethod
    def _ascend(delta, norm_type, adv_lr, adv_max_norm):
        """One projected ascent step on a perturbation (:2793-2850): + adv_lr * grad / |grad| per first-index slice, then
        scaled back into the adv_max_norm ball."""
        g = delta.grad.clone().detach()
        n0 = g.size(0)
        shape = (-1,) + (1,) * (g.dim() - 1)
        if norm_type == 'l2':
            denorm = torch.clamp(torch.norm(g.reshape(n0, -1), dim=1), min=1e-8).view(shape)
            delta = (delta + adv_lr * g / denorm).detach()
            if adv_max_norm > 0:
                dn = torch.norm(delta.reshape(n0, -1).float(), p=2, dim=1).detach()
                exceed = (dn > adv_max_norm).to(delta)
                delta = (delta * (adv_max_norm / dn * exceed + (1 - exceed)).view(shape)).detach()
        elif norm_type == 'linf':
            denorm = torch.clamp(torch.norm(g.reshape(n0, -1), dim=1, p=float('inf')), min=1e-8).view(shape)
            delta = (delta + adv_lr * g / denorm).detach()
            if adv_max_norm > 0:
                delta = torch.clamp(delta, -adv_max_norm, adv_max_norm).detach()
        else:
            raise NotImplementedError('Norm type {} not specified.'.format(norm_type))
        return delta

    def update_freelb_delta(self, model, delta, embeds_init, input_ids, norm_type='l2', adv_lr=1e-3, adv_max_norm=1e-2):
        """:2793-2827: ascent step on the text perturbation; the embeddings are looked up again (the step before moved them)."""
        return _unwrap(model).embeddings(input_ids), self._ascend(delta, norm_type, adv_lr, adv_max_norm)

    def update_image_freelb_delta(self, image_embeds, delta, norm_type='l2', adv_lr=1e-3, adv_max_norm=1e-2):
        """:2829-2851."""
        return self._ascend(delta, norm_type, adv_lr, adv_max_norm)

    def free_optimize(self, loss):
        """:2755-2791.  Follows the reference's AMP branch (the configuration its README runs): with
        ``accumulate_gradients`` > 1 the adversarial passes accumulate like any other step and only boundary iterations
        clip and step (its amp == -1 branch would step on every pass)."""
        self.optimize(loss)

    def _freelb_rel_step(self, batches, dataset, task_name, lambda_coeff):
        """freelb_t2i_step / freelb_i2t_step (:2021-2224, identical up to the statistics key): three passes over one batch, each
        with the word embeddings and the region features perturbed (``text_embed=`` / ``x_img +``), the loss / 3 optimised at
        once, and between passes one normalised ascent step of both perturbations along their gradients."""
        assert lambda_coeff >= 0
        if lambda_coeff == 0:
            return None
        params = self.params
        model = self.model
        model.train()
        (x1, len1, _lang_p), visual = batches[0], batches[1]
        img, img_mask, img_loc, _obj_labels, pos_labels, _img_ids = visual
        assert getattr(_unwrap(model), 'ddp_hook', None) is None or not params.multi_gpu, \
            'FreeLB steps are single-GPU in this build (the dense word-embedding gradient of text_embed has no reducer bucket)'
        self._dp_plan(False)
        img_len = img_mask.sum(dim=1)
        x_img, img_loc = img.transpose(0, 1), img_loc.transpose(0, 1)
        x1, len1, x_img, img_loc, img_len = to_cuda(x1, len1, x_img, img_loc, img_len)
        embeds_init, delta = self.deal_freelb_delta(model, x1.transpose(0, 1), len1)
        image_delta = self.deal_image_freelb_delta(x_img)
        adv_steps, tb_loss = 3, 0.0
        for astep in range(adv_steps):
            delta.requires_grad_()
            text_imb = delta + embeds_init
            image_delta.requires_grad_()
            img_imb = x_img + image_delta
            enc = model('jointfwd', x=x1, lengths=len1, x_img=img_imb, lengths_img=img_len, causal=False, langs=None,
                        image_loc=img_loc, refine_image=params.refine_image, text_embed=text_imb)
            enc = enc.transpose(0, 1)
            relation_scores = model('predict', tensor=enc, is_relation=True)
            loss = self._itm_loss(relation_scores, pos_labels) / (1.0 * adv_steps)
            # the word-embedding gradient of these passes arrives through autograd (text_embed = delta + Emb[x]), not through
            # the encoder's own backward: tell the arena that the matrix is part of this step
            _unwrap(model).arena().touch('embeddings.weight')
            self.free_optimize(loss)
            tb_loss = tb_loss + loss.detach()
            if astep == adv_steps - 1:
                break
            embeds_init, delta = self.update_freelb_delta(model, delta, embeds_init, x1.transpose(0, 1))
            image_delta = self.update_image_freelb_delta(x_img, image_delta)
        self._stat('FRLB-%s-%s' % (task_name, dataset), tb_loss)
        bs = len1.size(0)
        self.n_sentences += params.batch_size
        self.stats['processed_s'] += bs
        self.stats['processed_w'] += bs * enc.size(1)
        return tb_loss

    def get_ic_output(self, params, model, x2, len2, x1, len1, img_loc, langs, langs_img, pred_mask, y, adv_text_embed):
        """:2737-2753: the captioning pass with (optionally) caller-made word rows on the decoder side."""
        enc1 = model('crossfwd', stream_='img', x=x1, lengths=len1, langs=langs_img, causal=False, image_loc=img_loc,
                     refine_image=getattr(params, 'refine_image', False)).transpose(0, 1)
        dec2 = model('crossfwd', stream_='text', x=x2, lengths=len2, langs=langs, causal=True, src_enc=enc1, src_len=len1,
                     text_embed=adv_text_embed)
        _, loss = model('predict', tensor=dec2, pred_mask=pred_mask, y=y, get_scores=False)
        return loss, dec2

    def free_lb_ic_step(self, dataset='coco', input_stream='img', lambda_coeff=1):
        """FreeLB captioning step (xtrainer.py:2853-2962; train_x.py:454-455): three captioning passes over one batch with the
        caption's word embeddings (``params.free_text``) and / or the region features (``params.free_img``) perturbed, every
        pass an optimizer step on loss / 3, one normalised ascent step on the perturbations in between."""
        assert lambda_coeff >= 0
        if lambda_coeff == 0:
            return None
        params = self.params
        model = self.model
        model.train()
        assert getattr(_unwrap(model), 'ddp_hook', None) is None or not params.multi_gpu, 'FreeLB steps are single-GPU in this build'
        self._dp_plan(True, expect=('mlm',))
        (x2, len2), (x1, x1_mask, img_loc, _img_id) = self.get_batch('txt2img', dataset, input_stream)
        ft = getattr(params, 'ft_lgs', None) or []
        lang_id = params.lang2id[ft[0]] if len(ft) > 0 else params.lang2id['en']
        langs = x2.clone().fill_(lang_id)
        alen = torch.arange(int(len2.max()), dtype=torch.long, device=len2.device)
        pred_mask = alen[:, None] < len2[None] - 1
        y = x2[1:].masked_select(pred_mask[:-1])
        n_words = int((len2 - 1).sum())
        assert len(y) == n_words
        len1 = x1_mask.sum(dim=1)
        x1 = x1.transpose(0, 1)
        img_loc = img_loc.transpose(0, 1)
        langs_img = x1_mask.transpose(0, 1).clone().long().fill_(lang_id)
        x1, len1, img_loc, x2, len2, y, langs, langs_img, pred_mask = to_cuda(x1, len1, img_loc, x2, len2, y, langs, langs_img, pred_mask)
        x1 = x1.contiguous()
        free_text, free_img = getattr(params, 'free_text', False), getattr(params, 'free_img', False)
        if free_text:
            embeds_init, delta = self.deal_freelb_delta(model, x2.transpose(0, 1), len2)
        if free_img:
            image_delta = self.deal_image_freelb_delta(x1)
        adv_steps, tb_loss = 3, 0.0
        for astep in range(adv_steps):
            text_imb, img_imb = None, x1
            if free_text:
                delta.requires_grad_()
                text_imb = delta + embeds_init
                _unwrap(model).arena().touch('embeddings.weight')
            if free_img:
                image_delta.requires_grad_()
                img_imb = x1 + image_delta
            loss, _dec = self.get_ic_output(params, model, x2, len2, img_imb, len1, img_loc, langs, langs_img, pred_mask, y, text_imb)
            loss = loss / (1.0 * adv_steps)
            self.free_optimize(loss)
            tb_loss = tb_loss + loss.detach()
            if astep == adv_steps - 1:
                break
            if free_text:
                embeds_init, delta = self.update_freelb_delta(model, delta, embeds_init, x2.transpose(0, 1))
            if free_img:
                image_delta = self.update_image_freelb_delta(x1, image_delta)
        self._stat('FRLB-IC-%s-%s' % (dataset, input_stream), tb_loss)
        self.n_sentences += params.batch_size
        self.stats['processed_s'] += len2.size(0)
        self.stats['processed_w'] += n_words
        return tb_loss

    def freelb_t2i_step(self, batches, dataset='coco', lambda_coeff=1):
        return self._freelb_rel_step(batches, dataset, 't2i', lambda_coeff)

    def freelb_i2t_step(self, batches, dataset='coco', lambda_coeff=1):
        return self._freelb_rel_step(batches, dataset, 'i2t', lambda_coeff)

    def pretrain_under_step(self, _batch, dataset='coco', task_name='t2i', lang2='en', lambda_coeff_rel=1,
                            lambda_coeff_mlm=1, lambda_coeff_mrm=1, lambda_coeff_mrfr=1):
        """xtrainer.py:2234-2402: MLM + MRM + MRFR + ITM (+ CLCM on the i2t task) on one joint batch."""
        params = self.params
        model = self.model
        model.train()
        if task_name == 't2i':
            (x1, len1, x1_labels), (img, img_mask, img_loc, obj_labels, pos_labels, ori_att_feats, img_ids) = _batch
        else:
            (x1, len1, x1_labels), (x2, len2), (clcm_labels, img, img_mask, img_loc, obj_labels, pos_labels,
                                                ori_att_feats, img_ids) = _batch
        img_len = img_mask.sum(dim=1)
        x_img = img.transpose(0, 1)
        img_loc = img_loc.transpose(0, 1)
        y_text, pred_mask_text = self.get_mask_(x1, x1_labels)
        mlm_on, mrm_on, mrfr_on = (len(getattr(params, k)) > 0 for k in ('cross_mlm_steps', 'cross_mrm_steps', 'cross_mrfr_steps'))
        self._dp_plan(mlm_on, expect=[h for h, on in (('mlm', mlm_on), ('mrm', mrm_on), ('mrfr', mrfr_on)) if on])
        has_mlm = mlm_on and int(y_text.numel()) > 0
        x1, len1, x_img, img_loc, img_len, y_text, pred_mask_text = to_cuda(
            x1, len1, x_img, img_loc, img_len, y_text, pred_mask_text)

        encoder_outputs = model('jointfwd', x=x1, lengths=len1, x_img=x_img, lengths_img=img_len, causal=False,
                                langs=None, image_loc=img_loc, refine_image=params.refine_image)
        total_loss = None
        R = x_img.shape[0]
        _text_out = encoder_outputs[R:]
        if has_mlm:
            _, loss = model('predict', tensor=_text_out, pred_mask=pred_mask_text, y=y_text, get_scores=False)
            self._stat('CMLM-%s' % dataset, loss)
            total_loss = _add_loss(total_loss, lambda_coeff_mlm, loss)
        _img_out = encoder_outputs[:R].transpose(0, 1)          # (B, R, d), xtrainer.py:2288-2289
        has_masked_region = bool((obj_labels != -1).any()) if torch.is_tensor(obj_labels) else False   # host tensor
        if mrm_on and has_masked_region:          # xtrainer.py:2320-2328
            _, loss = model('predict', tensor=_img_out, pred_mask=None, y=obj_labels.reshape(-1), get_scores=False, is_obj=True)
            self._stat('MRM-%s' % dataset, loss)
            total_loss = _add_loss(total_loss, lambda_coeff_mrm, loss)
        if mrfr_on and has_masked_region:         # xtrainer.py:2330-2352
            from . import functional as Fn
            loss = Fn.mrfr_head(_unwrap(model), _img_out, obj_labels, ori_att_feats)
            self._stat('MRFR-%s' % dataset, loss)
            total_loss = _add_loss(total_loss, lambda_coeff_mrfr, loss)
        relation_scores = model('predict', tensor=encoder_outputs.transpose(0, 1), is_relation=True)
        loss = self._itm_loss(relation_scores, pos_labels)
        self._stat('%s-%s' % (task_name, dataset), loss)
        total_loss = _add_loss(total_loss, lambda_coeff_rel, loss)

        if task_name == 'i2t' and len(params.cross_clcm_steps) > 0:        # xtrainer.py:2379-2393
            x2c, len2c = to_cuda(x2, len2)
            encoder_outputs2 = model('jointfwd', x=x2c, lengths=len2c, x_img=x_img, lengths_img=img_len, causal=False,
                                     langs=None, image_loc=img_loc, refine_image=params.refine_image)
            relation_scores2 = model('predict', tensor=encoder_outputs2.transpose(0, 1), is_clcm=True)
            target2 = torch.as_tensor(clcm_labels).reshape(-1).to(device=relation_scores2.device, dtype=torch.float32)
            loss = F.binary_cross_entropy_with_logits(relation_scores2.view(-1).float(), target2)
            self._stat('CLCM-%s' % dataset, loss)
            total_loss = _add_loss(total_loss, 1, loss)

        self.optimize(total_loss)
        self.n_sentences += params.batch_size
        self.stats['processed_s'] += len1.size(0)
        self._pending_w.append(len1.sum())
        return total_loss

    def _rel_step(self, batches, dataset, task_name, lambda_coeff):
        """t2i_step / i2t_step (xtrainer.py:1888-2018): jointfwd -> relation scores -> CE/BCE -> optimize.
        ``batches`` is what retrieval_collate emits: ((x1, len1, lang_p), (img, img_mask, img_loc, obj_labels,
        pos_labels, img_ids)); every dataset item contributes sample_n sequences.  The language ids only feed the
        ``langs`` argument, which jointfwd ignores (transformer.py:937-938)."""
        assert lambda_coeff >= 0
        if lambda_coeff == 0:
            return None
        params = self.params
        model = self.model
        model.train()
        text, visual = batches[0], batches[1]
        x1, len1 = text[0], text[1]
        if len(visual) == 6:
            img, img_mask, img_loc, _obj_labels, pos_labels, _img_ids = visual
        else:                                      # (img, img_mask, img_loc, pos_labels): hand-built batches
            img, img_mask, img_loc, pos_labels = visual
        self._dp_plan(False)
        img_len = img_mask.sum(dim=1)
        x_img, img_loc = img.transpose(0, 1), img_loc.transpose(0, 1)
        x1, len1, x_img, img_loc, img_len = to_cuda(x1, len1, x_img, img_loc, img_len)
        enc = model('jointfwd', x=x1, lengths=len1, x_img=x_img, lengths_img=img_len, causal=False, langs=None,
                    image_loc=img_loc, refine_image=params.refine_image)
        enc = enc.transpose(0, 1)
        relation_scores = model('predict', tensor=enc, is_relation=True)
        loss = self._itm_loss(relation_scores, pos_labels)
        self._stat('%s-%s' % (task_name, dataset), loss)
        self.optimize(lambda_coeff * loss)
        bs = len1.size(0)
        self.n_sentences += params.batch_size
        self.stats['processed_s'] += bs
        self.stats['processed_w'] += bs * enc.size(1)
        return loss

    def t2i_step(self, batches, dataset='coco', lambda_coeff=1):
        return self._rel_step(batches, dataset, 't2i', lambda_coeff)

    def i2t_step(self, batches, dataset='coco', lambda_coeff=1):
        return self._rel_step(batches, dataset, 'i2t', lambda_coeff)

    def slide_step(self, dataset='slide', input_stream='img', lambda_coeff=1):
        """Sliding-window matching step (xtrainer.py:2649-2698) on a ``slide_collate`` batch
        ``(x2, len2), (x1, x1_mask, img_loc, img_id), labels``: jointfwd -> relation score per sequence -> BCE against the
        0 / 1 labels (kept on the device here; the reference moves the scores to the host first)."""
        assert lambda_coeff >= 0
        if lambda_coeff == 0:
            return None
        params = self.params
        model = self.model
        model.train()
        (x2, len2), (x1, x1_mask, img_loc, _img_id), labels = self.get_batch('slide2img', dataset, input_stream)
        self._dp_plan(False)
        img_len = x1_mask.sum(dim=1)
        x_img, img_loc = x1.transpose(0, 1), img_loc.transpose(0, 1)
        target = torch.as_tensor(np.asarray(labels, dtype='float32')).view(-1)
        x2, len2, x_img, img_loc, img_len, target = to_cuda(x2, len2, x_img, img_loc, img_len, target)
        enc = model('jointfwd', x=x2, lengths=len2, x_img=x_img, lengths_img=img_len, causal=False, langs=None,
                    image_loc=img_loc, refine_image=params.refine_image)
        relation_scores = model('predict', tensor=enc.transpose(0, 1), is_relation=True)
        loss = F.binary_cross_entropy_with_logits(relation_scores.float().view(-1), target)
        self._stat('SLIDE-%s' % input_stream, loss)
        self.optimize(lambda_coeff * loss)
        self.n_sentences += params.batch_size
        self.stats['processed_s'] += len2.size(0)
        self.stats['processed_w'] += int((len2 - 1).sum())
        return loss.detach()
