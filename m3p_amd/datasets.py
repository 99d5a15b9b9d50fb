"""Token-stream dataset of the text-only MLM step: the data format on the input side of ``Trainer.mlm_step``
(``data['mono_stream'][lang]['train']``; M3P/src/data/dataset_pretrain.py:787-890 ``StreamDataset``).  Host-side numpy
only - the GPU path starts at ``mlm_step_on_batch``.

A monolingual corpus arrives as one long vector of word ids in which every sentence ends with EOS, plus the (start, end)
positions of the sentences.  The stream is cut into ``batch_size`` parallel lanes and served ``bptt`` rows at a time:

* the vector is left-padded with EOS to ``n_batches * bptt * batch_size`` ids and laid out lane-major, so lane ``b`` is
  the contiguous slice ``[b * n_batches * bptt, (b + 1) * n_batches * bptt)`` of the padded stream (:809-813);
* one EOS row is put in front (:812-813), so the matrix has ``n_batches * bptt + 1`` rows; batch ``i`` is rows
  ``[i * bptt, (i + 1) * bptt)`` - the last row is never served;
* every batch reports the same lengths, ``bptt`` for each lane (:830);
* language ids, if given, go through the same lane layout *without* the extra first row (:819-821 builds the
  (n + 1)-row matrix and then replaces it by the n-row one) - kept as the reference serves it;
* a shuffled epoch walks a ``RandomState(seed + number of epochs started on this rank)`` permutation of the batch
  indices; ``loaded[rank]`` counts the batches handed out in each epoch so that a reloaded run (``reload_check``)
  skips the ones its interrupted epoch had already served (:866-886)."""
import math
from logging import getLogger

import numpy as np
import torch

logger = getLogger()


class StreamDataset(object):
    def __init__(self, sent, pos, params, langs=()):
        self.params = params
        self.bptt = bptt = params.bptt
        bs = params.batch_size
        self.eos = params.eos_index
        self.n_gpu_per_node = getattr(params, 'n_gpu_per_node', 1)
        self.local_rank = getattr(params, 'local_rank', 0)
        # one position pair per sentence, each ending on an EOS
        assert len(pos) == (sent == self.eos).sum()
        assert len(pos) == (sent[pos[:, 1]] == self.eos).sum()

        self.n_tokens = n_tokens = len(sent)
        self.n_batches = math.ceil(n_tokens / (bs * bptt))
        rows = self.n_batches * bptt

        def lanes(values, fill):
            flat = np.full(rows * bs, fill, dtype=sent.dtype)
            flat[rows * bs - n_tokens:] = values
            return flat.reshape(bs, rows).T

        self.data = np.full((rows + 1, bs), self.eos, dtype=sent.dtype)
        self.data[1:] = lanes(sent, self.eos)
        self.has_lan = len(langs) != 0
        if self.has_lan:
            self.langs = lanes(langs, params.lang2id['en'])
        self.n_sentences = len(pos)
        self.loaded = {r: [] for r in range(self.n_gpu_per_node)}
        self.reload = False
        self.lengths = torch.LongTensor(bs).fill_(bptt)

    def __len__(self):
        return self.n_sentences

    def reload_check(self, loaded):
        logger.info('reload records [%s]' % ','.join(str(v) for v in loaded[self.local_rank]))
        self.loaded = loaded
        self.reload = True

    def select_data(self, a, b):
        """Keep batches [a, b) only (:846-864)."""
        if not (0 <= a < b <= self.n_batches):
            logger.warning('Invalid split values: %i %i - %i' % (a, b, self.n_batches))
            return
        logger.info('Selecting batches from %i to %i ...' % (a, b))
        self.data = np.copy(self.data[a * self.bptt:b * self.bptt])
        if self.has_lan:
            self.langs = np.copy(self.langs[a * self.bptt:b * self.bptt])
        self.n_batches = b - a
        self.n_sentences = int((self.data == self.eos).sum())

    def get_iterator(self, shuffle, subsample=1, seed=0):
        """Yields ``(x (bptt, batch_size) int64, lengths[, langs])`` (:866-890)."""
        mine = self.loaded[self.local_rank]
        if not self.reload:
            mine.append(0)
        n = self.n_batches // subsample
        if shuffle:
            if seed == 0:
                seed = np.random.randint(1, 1e6)
            seed += len(mine)
            logger.warning('GPU %s shuffled with seed %s' % (self.local_rank, seed))
            order = np.random.RandomState(seed).permutation(n)
        else:
            order = range(n)
        for k, i in enumerate(order):
            if shuffle and self.reload and k < mine[-1]:
                continue                          # served before the checkpoint this epoch was resumed from
            rows = slice(self.bptt * i, self.bptt * (i + 1))
            mine[-1] += 1
            self.reload = False
            x = torch.from_numpy(self.data[rows].astype(np.int64))
            if self.has_lan:
                yield x, self.lengths, torch.from_numpy(self.langs[rows].astype(np.int64))
            else:
                yield x, self.lengths
