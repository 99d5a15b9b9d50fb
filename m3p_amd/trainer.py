"""Trainer surface of the hot path: the parts of ``Trainer`` / ``XTrainer``
(M3P/src/xtrainer.py:35-826, 1128-2961) that the MLM + ITM pre-training step and the ITM
fine-tune step execute, driving the MI355X model.

Kept: constructor wiring (parameters, optimizer, DDP), ``optimize`` (:205-243), ``iter`` /
``print_stats`` (:245-289, the reference's own sent/s meter), ``get_mask_`` (:2226-2232),
``pretrain_under_step`` (:2234-2402), ``t2i_step`` / ``i2t_step`` loss arithmetic
(:1888-2018), ``mlm_step`` (:734-770) on a caller-supplied batch, checkpoint save/reload of
model + optimizer (:511-599).

Changed on purpose (SURVEY.md §7 "host side clean"): no per-step host syncs — the NaN check
(:210), the ``.cpu()`` ITM loss (:2367-2370) and the ``loss.item()`` statistics
(:2317, :2374) stay on the device and are only read when ``print_stats`` prints;
clip + Adam + zero_grad are one fused kernel pass; DDP is our bucketed reducer.
"""
import os
import time
from collections import OrderedDict
from logging import getLogger

import numpy as np
import torch
import torch.nn.functional as F

from .distributed import DataParallel
from .optim import get_optimizer

logger = getLogger()


def to_cuda(*args):
    """M3P/src/utils.py:233-237."""
    return [None if x is None else x.cuda(non_blocking=True) for x in args]


def batch_sentences_v2(sentences, lm_labels=None, pad_index=1, bos_index=0, eos_index=2):
    """Collate of xtrainer.py:855-880: (slen, n) int64 with BOS first, EOS last, PAD after;
    labels -1 where nothing is predicted."""
    lengths = torch.LongTensor([len(s) + 2 for s in sentences])
    slen, n = int(lengths.max()), len(sentences)
    sent = torch.full((slen, n), pad_index, dtype=torch.long)
    labels = torch.full((slen, n), -1, dtype=torch.long) if lm_labels is not None else None
    sent[0] = bos_index
    for i, s in enumerate(sentences):
        li = int(lengths[i])
        if li > 2:
            sent[1:li - 1, i] = torch.from_numpy(np.asarray(s).astype(np.int64))
            if lm_labels is not None:
                labels[1:li - 1, i] = torch.from_numpy(np.asarray(lm_labels[i]).astype(np.int64))
        sent[li - 1, i] = eos_index
    if lm_labels is not None:
        return sent, lengths, labels
    return sent, lengths


class Trainer(object):
    MODEL_NAMES = ['model']

    def __init__(self, data, params):
        self.epoch_size = params.epoch_size
        self.params = params
        self.data = data
        self.stopping_criterion = None
        self.best_stopping_criterion = None
        self.iterators = {}
        self.set_parameters()
        assert params.amp >= 1 or not params.fp16
        # bf16 compute with fp32 master weights is the only precision mode of this build;
        # `amp`/`fp16` are accepted for flag compatibility (no loss scaling: bf16 keeps fp32's range)
        self.set_optimizers()
        if getattr(params, 'multi_gpu', False):
            logger.info('Using m3p_amd.distributed.DataParallel (bucketed RCCL all-reduce) ...')
            for name in self.MODEL_NAMES:
                wrapped = DataParallel(getattr(self, name))
                setattr(self, name, wrapped)
                for opt in self.optimizers.values():
                    opt.grad_scale = 1.0 / wrapped.world
        self.metrics = []
        self.best_metrics = {}
        self.epoch = 0
        self.n_iter = 0
        self.n_total_iter = 0
        self.n_sentences = 0
        self.stats = OrderedDict([('processed_s', 0), ('processed_w', 0)])
        self.last_time = time.time()
        self._pending_w = []

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

    def optimize(self, loss):
        """xtrainer.py:205-243 without the host round trips: backward -> (bucketed
        all-reduce overlapped with it) -> global-norm clip -> Adam -> zero_grad, the last
        three as one fused pass over the arenas."""
        params = self.params
        optimizers = list(self.optimizers.values())
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
        for opt in optimizers:
            if params.clip_grad_norm > 0:
                opt.clip_grad_norm(params.clip_grad_norm)
            opt.step()

    def iter(self):
        """xtrainer.py:245-252."""
        self.n_iter += 1
        self.n_total_iter += 1
        self.print_stats()

    def print_stats(self):
        """xtrainer.py:254-289 (sent/s = processed_s / elapsed is the throughput metric)."""
        if self.n_iter % 5 != 0:
            return
        if self._pending_w:
            self.stats['processed_w'] += int(torch.stack(self._pending_w).sum().item())
            self._pending_w = []
        s_iter = '%7i - ' % self.n_iter
        parts = []
        for k, v in self.stats.items():
            if type(v) is list and len(v) > 0:
                vals = [float(x) for x in (torch.stack([t.float() for t in v]).tolist() if torch.is_tensor(v[0]) else v)]
                parts.append('{}: {:7.4f}'.format(k, np.mean(vals)))
                del v[:]
        s_stat = ' || '.join(parts)
        s_lr = ' - '
        for k, v in self.optimizers.items():
            s_lr = s_lr + (' - %s LR: ' % k) + ' / '.join('{:.4e}'.format(group['lr']) for group in v.param_groups)
        new_time = time.time()
        diff = new_time - self.last_time
        s_speed = '{:7.2f} sent/s - {:8.2f} words/s - '.format(self.stats['processed_s'] * 1.0 / diff,
                                                              self.stats['processed_w'] * 1.0 / diff)
        self.stats['processed_s'] = 0
        self.stats['processed_w'] = 0
        self.last_time = new_time
        logger.info(s_iter + s_speed + s_stat + s_lr)

    # ---- checkpoints (xtrainer.py:511-599): same dict layout / key names
    def save_model(self, name):
        path = os.path.join(self.params.dump_path, '%s.pth' % name)
        data = {}
        for n in self.MODEL_NAMES:
            m = getattr(self, n)
            m = m.module if isinstance(m, DataParallel) else m
            data[n] = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        data['params'] = {k: v for k, v in self.params.__dict__.items() if not torch.is_tensor(v)}
        torch.save(data, path)
        return path

    def save_checkpoint(self, name='checkpoint'):
        path = os.path.join(self.params.dump_path, '%s.pth' % name)
        data = {'epoch': self.epoch, 'n_total_iter': self.n_total_iter, 'best_metrics': self.best_metrics,
                'best_stopping_criterion': self.best_stopping_criterion}
        for n in self.MODEL_NAMES:
            m = getattr(self, n)
            m = m.module if isinstance(m, DataParallel) else m
            data[n] = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        for n, opt in self.optimizers.items():
            data[n + '_optimizer'] = {'param_groups': [{k: v for k, v in g.items() if k != 'params'}
                                                       for g in opt.param_groups]}
        data['params'] = {k: v for k, v in self.params.__dict__.items() if not torch.is_tensor(v)}
        torch.save(data, path)
        return path

    def reload_checkpoint(self, path):
        """Restores weights and, like the reference (:586-592), only num_updates / lr of the optimizer."""
        data = torch.load(path, map_location='cpu', weights_only=False)
        for n in self.MODEL_NAMES:
            m = getattr(self, n)
            m = m.module if isinstance(m, DataParallel) else m
            sd = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in data[n].items()}
            m.load_state_dict(sd)
        for n, opt in self.optimizers.items():
            for gid, g in enumerate(opt.param_groups):
                saved = data[n + '_optimizer']['param_groups'][gid]
                if 'num_updates' in saved:
                    g['num_updates'] = saved['num_updates']
                    g['lr'] = opt.get_lr_for_step(g['num_updates'])
        self.epoch = data['epoch'] + 1 if 'epoch' in data else self.epoch
        self.n_total_iter = data.get('n_total_iter', self.n_total_iter)


class XTrainer(Trainer):
    def __init__(self, model, data, params):
        """xtrainer.py:1130-1146."""
        self.MODEL_NAMES = ['model']
        self.model = model
        self.data = data
        self.params = params
        super().__init__(data, params)

    # ------------------------------------------------------------------ masks
    def get_mask_(self, x, _labels):
        """xtrainer.py:2226-2232: mask = labels != -1; targets = labels[labels > 0]."""
        pred_mask = (_labels != -1)
        y = _labels[_labels > 0]
        return y, pred_mask

    def _itm_loss(self, relation_scores, pos_labels):
        """xtrainer.py:2357-2372 kept on the device: CE over groups of sample_n + BCE vs one-hot."""
        params = self.params
        dev = relation_scores.device
        pos = torch.as_tensor(np.asarray(pos_labels), dtype=torch.long).to(dev)
        onehot = F.one_hot(pos, params.sample_n).float().view(-1)
        scores = relation_scores.float()
        loss = 0
        if params.multi_cls_loss_weight != 0:
            loss = loss + params.multi_cls_loss_weight * F.cross_entropy(scores.view(-1, params.sample_n), pos)
        if params.bin_cls_loss_weight != 0:
            loss = loss + params.bin_cls_loss_weight * F.binary_cross_entropy_with_logits(scores.view(-1), onehot)
        return loss

    # ------------------------------------------------------------------ hot steps
    def pretrain_under_step(self, _batch, dataset='coco', task_name='t2i', lang2='en', lambda_coeff_rel=1,
                            lambda_coeff_mlm=1, lambda_coeff_mrm=1, lambda_coeff_mrfr=1):
        """xtrainer.py:2234-2402 for the MLM (+ITM) objective."""
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
        has_mlm = len(params.cross_mlm_steps) > 0 and int(y_text.numel()) > 0
        x1, len1, x_img, img_loc, img_len, y_text, pred_mask_text = to_cuda(
            x1, len1, x_img, img_loc, img_len, y_text, pred_mask_text)

        encoder_outputs = model('jointfwd', x=x1, lengths=len1, x_img=x_img, lengths_img=img_len, causal=False,
                                langs=None, image_loc=img_loc, refine_image=params.refine_image)
        total_loss = 0
        R = x_img.shape[0]
        _text_out = encoder_outputs[R:]
        if has_mlm:
            _, loss = model('predict', tensor=_text_out, pred_mask=pred_mask_text, y=y_text, get_scores=False)
            self._stat('CMLM-%s' % dataset, loss)
            total_loss = total_loss + lambda_coeff_mlm * loss
        _img_out = encoder_outputs[:R].transpose(0, 1)          # (B, R, d), xtrainer.py:2288-2289
        has_masked_region = bool((obj_labels != -1).any()) if torch.is_tensor(obj_labels) else False   # host tensor
        if len(params.cross_mrm_steps) > 0 and has_masked_region:          # xtrainer.py:2320-2328
            _, loss = model('predict', tensor=_img_out, pred_mask=None, y=obj_labels.reshape(-1), get_scores=False, is_obj=True)
            self._stat('MRM-%s' % dataset, loss)
            total_loss = total_loss + lambda_coeff_mrm * loss
        if len(params.cross_mrfr_steps) > 0 and has_masked_region:         # xtrainer.py:2330-2352
            from . import functional as Fn
            loss = Fn.mrfr_head(model.module if hasattr(model, 'module') else model, _img_out, obj_labels, ori_att_feats)
            self._stat('MRFR-%s' % dataset, loss)
            total_loss = total_loss + lambda_coeff_mrfr * loss
        relation_scores = model('predict', tensor=encoder_outputs.transpose(0, 1), is_relation=True)
        loss = self._itm_loss(relation_scores, pos_labels)
        self._stat('%s-%s' % (task_name, dataset), loss)
        total_loss = total_loss + lambda_coeff_rel * loss

        if task_name == 'i2t' and len(params.cross_clcm_steps) > 0:        # xtrainer.py:2379-2393
            x2c, len2c = to_cuda(x2, len2)
            encoder_outputs2 = model('jointfwd', x=x2c, lengths=len2c, x_img=x_img, lengths_img=img_len, causal=False,
                                     langs=None, image_loc=img_loc, refine_image=params.refine_image)
            relation_scores2 = model('predict', tensor=encoder_outputs2.transpose(0, 1), is_clcm=True)
            target2 = torch.as_tensor(clcm_labels).reshape(-1).to(device=relation_scores2.device, dtype=torch.float32)
            loss = F.binary_cross_entropy_with_logits(relation_scores2.view(-1).float(), target2)
            self._stat('CLCM-%s' % dataset, loss)
            total_loss = total_loss + loss

        self.optimize(total_loss)
        self.n_sentences += params.batch_size
        self.stats['processed_s'] += len1.size(0)
        self._pending_w.append(len1.sum())
        return total_loss

    def _rel_step(self, _batch, dataset, task_name, lambda_coeff):
        """t2i_step / i2t_step loss path (xtrainer.py:1888-2018): jointfwd -> relation scores ->
        CE/BCE -> optimize; each dataset item contributes sample_n sequences."""
        params = self.params
        model = self.model
        model.train()
        (x1, len1), (img, img_mask, img_loc, pos_labels) = _batch[:2]
        img_len = img_mask.sum(dim=1)
        x_img, img_loc = img.transpose(0, 1), img_loc.transpose(0, 1)
        x1, len1, x_img, img_loc, img_len = to_cuda(x1, len1, x_img, img_loc, img_len)
        enc = model('jointfwd', x=x1, lengths=len1, x_img=x_img, lengths_img=img_len, causal=False, langs=None,
                    image_loc=img_loc, refine_image=params.refine_image)
        relation_scores = model('predict', tensor=enc.transpose(0, 1), is_relation=True)
        loss = self._itm_loss(relation_scores, pos_labels)
        self._stat('%s-%s' % (task_name, dataset), loss)
        self.optimize(lambda_coeff * loss)
        self.n_sentences += params.batch_size
        self.stats['processed_s'] += len1.size(0)
        self._pending_w.append(len1.sum())
        return loss

    def t2i_step(self, _batch, dataset='coco', lambda_coeff=1):
        return self._rel_step(_batch, dataset, 't2i', lambda_coeff)

    def i2t_step(self, _batch, dataset='coco', lambda_coeff=1):
        return self._rel_step(_batch, dataset, 'i2t', lambda_coeff)

    def mlm_step_on_batch(self, x, lengths, pred_mask, y, lang='en', lambda_coeff=1):
        """Loss path of Trainer.mlm_step (xtrainer.py:734-770) on an already masked batch
        (mask_out :385-434 is host-side numpy RNG and stays with the data layer)."""
        model = self.model
        model.train()
        x, lengths, pred_mask, y = to_cuda(x, lengths, pred_mask, y)
        tensor = model('crossfwd', stream_='text', x=x, lengths=lengths, positions=None, langs=None, causal=False)
        _, loss = model('predict', tensor=tensor, pred_mask=pred_mask, y=y, get_scores=False)
        self._stat('MLM-%s' % lang, loss)
        self.optimize(lambda_coeff * loss)
        self.n_sentences += self.params.batch_size
        self.stats['processed_s'] += lengths.size(0)
        self._pending_w.append(pred_mask.sum())
        return loss
