"""Optimizers of the hot path: drop-in for ``M3P/src/optim.py`` (Adam :16-86,
AdamInverseSqrtWithWarmup :89-139, get_optimizer :211-270) with the per-tensor Python
loop replaced by fused HIP kernels over the model's flat arenas
(csrc/optim.hip: m3p_sumsq_f32 + m3p_adam_step = clip + Adam + bf16 refresh + zero_grad in
one pass over p, g, m, v).

Semantics kept from the reference:
  * state is allocated eagerly in the constructor; ``state[p]`` holds ``step`` (int),
    ``exp_avg``, ``exp_avg_sq`` (here: views into flat moment arenas);
  * parameters that received no gradient in a step are skipped entirely (optim.py:55-56);
  * eps is added to sqrt(v) before the bias-corrected step size is applied;
    weight decay is the decoupled ``p -= wd * lr * p`` (:81-82);
  * AdamInverseSqrtWithWarmup is constructed with lr = warmup_init_lr and moves the lr
    *after* each step (:135-139); ``param_groups[*]['num_updates']`` is checkpoint-visible.
"""
import inspect
import math
import re

import torch
from torch import optim

import os

from . import ops

# developer switch for A/B runs: 0 = the fused Adam pass zeroes the vocabulary gradient like every other range (round 4)
_LAZY_VOCAB_ZERO = os.environ.get('M3P_LAZY_VOCAB_ZERO', '1') != '0'


class Adam(optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if not 0.0 <= lr:
            raise ValueError('Invalid learning rate: {}'.format(lr))
        if not 0.0 <= eps:
            raise ValueError('Invalid epsilon value: {}'.format(eps))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError('Invalid beta parameter at index 0: {}'.format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError('Invalid beta parameter at index 1: {}'.format(betas[1]))
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._arenas = {}        # id(arena) -> dict(arena, m, v)
        self._pending_clip = None
        self.grad_scale = 1.0    # 1/world_size under data parallelism (gradient averaging)
        for group in self.param_groups:
            for p in group['params']:
                state = self.state[p]
                state['step'] = 0
                tag = getattr(p, '_m3p_arena', None)
                if tag is not None:
                    arena, name = tag
                    ent = self._arenas.get(id(arena))
                    if ent is None:
                        ent = dict(arena=arena, m=torch.zeros_like(arena.master), v=torch.zeros_like(arena.master),
                                   gnorm=torch.zeros(1, dtype=torch.float64, device=arena.device))
                        self._arenas[id(arena)] = ent
                    o, cnt, shape = arena.offsets[name]
                    state['exp_avg'] = ent['m'][o:o + cnt].view(shape)
                    state['exp_avg_sq'] = ent['v'][o:o + cnt].view(shape)
                else:
                    state['exp_avg'] = torch.zeros_like(p.data)
                    state['exp_avg_sq'] = torch.zeros_like(p.data)

    # ------------------------------------------------------------------ helpers
    def _active_ranges(self, arena):
        """Contiguous arena ranges of the parameters touched by backward since the last
        zero_grad, split where the per-parameter step counts differ."""
        by_param = {}
        for group in self.param_groups:
            for p in group['params']:
                tag = getattr(p, '_m3p_arena', None)
                if tag is not None and tag[0] is arena:
                    by_param[tag[1]] = (p, group)
        ranges = []
        cur = None
        for name in arena.names:
            if name not in by_param or name not in arena.touched:
                cur = None
                continue
            p, group = by_param[name]
            o, cnt, _ = arena.offsets[name]
            end = o + (cnt + 63) // 64 * 64
            step = self.state[p]['step']
            # (the arena pads to 512 elements in front of a gradient bucket: a gap between two consecutive touched
            #  parameters is that padding - zeros in every arena, a no-op for Adam - and does not split the range)
            if cur is not None and cur['next'] == name and cur['step'] == step and cur['group'] is group:
                cur['end'] = end
                cur['params'].append(p)
            else:
                cur = dict(start=o, end=end, step=step, group=group, params=[p])
                ranges.append(cur)
            cur['next'] = self._next_name(arena, name)
        return ranges

    @staticmethod
    def _next_name(arena, name):
        order = getattr(arena, '_name_after', None)
        if order is None:
            order = arena._name_after = dict(zip(arena.names, arena.names[1:] + [None]))
        return order[name]

    @staticmethod
    def _hook(arena):
        return getattr(arena.model, 'ddp_hook', None)

    def _owned(self, arena, start, end):
        """[start, end) restricted to what this rank steps: all of it, or - under the sharded data-parallel exchange
        (distributed.py, zero1) - its intersections with this rank's bucket shards."""
        hook = self._hook(arena)
        return hook.owned(start, end) if hook is not None else [(start, end)]

    def clip_grad_norm(self, max_norm):
        """clip_grad_norm_(parameters, max_norm) of xtrainer.py:225, deferred: the global
        sum of squares is reduced on the device now, the scaling is applied inside the Adam
        kernel (no host sync, no extra pass over the gradients)."""
        for ent in self._arenas.values():
            arena = ent['arena']
            if arena.model.ddp_hook is not None:
                arena.model.ddp_hook.finish()
            ent['gnorm'].zero_()
            pieces = []
            for r in self._active_ranges(arena):
                pieces.extend(self._owned(arena, r['start'], r['end']))
            ops.sumsq_ranges(arena.grad, pieces, ent['gnorm'])      # (one launch however many shards this rank owns)
            if arena.model.ddp_hook is not None:
                arena.model.ddp_hook.all_reduce_scalar(ent['gnorm'])     # sharded gradients: the norm is the sum over ranks
        self._pending_clip = float(max_norm)

    def grad_norm(self):
        """Host value of the last reduced global norm (synchronises; for logging/tests)."""
        tot = sum(float(ent['gnorm'].item()) for ent in self._arenas.values())
        return math.sqrt(tot) * self.grad_scale

    def zero_grad(self, set_to_none=True):
        for ent in self._arenas.values():
            ent['arena'].zero_grad()
        for group in self.param_groups:
            for p in group['params']:
                if getattr(p, '_m3p_arena', None) is None:
                    p.grad = None

    # ------------------------------------------------------------------ step
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        max_norm = self._pending_clip or 0.0
        self._pending_clip = None
        for ent in self._arenas.values():
            arena = ent['arena']
            if arena.model.ddp_hook is not None:
                arena.model.ddp_hook.finish()
            active = self._active_ranges(arena)
            # this step's MLM head stored its weight gradient over the tied vocabulary matrix (it will again next step): that
            # range is not zeroed here - Arena.defer_vocab_zero, functional.py.  The range may span several active ranges
            # (per-parameter step counts differ after a step without an MLM head): every piece is cut at its borders.
            # Under data parallelism too (round 6: the wrapped step paid 0.3 ms for these zeros, profiles/r06_dp_overhead.txt):
            # the bucket collectives work on the arena in place after the store, the exchanged token rows enter through
            # Arena.g() (which zeroes a still-stale range first), and a sharded rank leaves its foreign shards of the range
            # un-zeroed as well (after_sharded_step(keep=...)) - the store covers them.
            lazy = None
            if getattr(arena, 'vocab_stored', False) and _LAZY_VOCAB_ZERO:
                v0, vc = arena.vocab_range()
                covered = sum(max(0, min(r['end'], v0 + vc) - max(r['start'], v0)) for r in active)
                if covered == vc:
                    lazy = (v0, v0 + vc)
                    arena.defer_vocab_zero()
            # every piece of every active range of a parameter group goes into ONE launch (ops.adam_step_ranges): a sharded rank
            # steps fifteen bucket shards, and fifteen launches streamed the same bytes 24 % slower than one
            batches = {}
            bump = []          # (parameter, new step count): applied only once every launch below has been accepted
            for r in active:
                group = r['group']
                beta1, beta2 = group['betas']
                step = r['step'] + 1
                bc1 = 1 - beta1 ** step
                bc2 = 1 - beta2 ** step
                step_size = group['lr'] * math.sqrt(bc2) / bc1
                pieces = batches.setdefault(id(group), (group, []))[1]
                for s, e in self._owned(arena, r['start'], r['end']):
                    if lazy is None or e <= lazy[0] or s >= lazy[1]:
                        pieces.append((s, e, step_size, True))
                    else:
                        cuts = sorted({s, e, min(max(lazy[0], s), e), min(max(lazy[1], s), e)})
                        pieces.extend((a, b, step_size, not (lazy[0] <= a and b <= lazy[1])) for a, b in zip(cuts, cuts[1:]) if b > a)
                bump.extend((p, step) for p in r['params'])
            for group, pieces in batches.values():
                # (the kernel takes pieces whose borders are multiples of four elements: a bad one is refused HERE, before any
                #  launch of this step has changed a parameter or a step counter)
                bad = [(a, b) for a, b, _, _ in pieces if (a | b) & 3]
                if bad:
                    raise ValueError('fused Adam: piece borders must be multiples of 4 elements, got %s' % bad[:3])
            for group, pieces in batches.values():
                beta1, beta2 = group['betas']
                ops.adam_step_ranges(arena.master, arena.grad, ent['m'], ent['v'], arena.w16, pieces, group['lr'], beta1, beta2,
                                     group['eps'], group['weight_decay'], gnorm_sq=ent['gnorm'] if max_norm > 0 else None,
                                     max_norm=max_norm, grad_scale=self.grad_scale)
            for p, step in bump:
                self.state[p]['step'] = step
            # sharded exchange: the other ranks' shards of the updated master come back through an all-gather that the
            # next forward waits for bucket by bucket (distributed.DataParallel.after_sharded_step)
            hook = arena.model.ddp_hook
            gathered = hook is not None and hook.after_sharded_step([(r['start'], r['end']) for r in active], keep=lazy)
            # untouched ranges may still hold stale values only if someone wrote them by hand
            arena.after_fused_step(copies_scheduled=gathered)
        # parameters outside any arena (never the case on the hot path) — reference loop
        for group in self.param_groups:
            for p in group['params']:
                if getattr(p, '_m3p_arena', None) is not None or p.grad is None:
                    continue
                grad = p.grad.data
                state = self.state[p]
                beta1, beta2 = group['betas']
                state['step'] += 1
                state['exp_avg'].mul_(beta1).add_(grad, alpha=1 - beta1)
                state['exp_avg_sq'].mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
                denom = state['exp_avg_sq'].sqrt().add_(group['eps'])
                bc1 = 1 - beta1 ** state['step']
                bc2 = 1 - beta2 ** state['step']
                step_size = group['lr'] * math.sqrt(bc2) / bc1
                if group['weight_decay'] != 0:
                    p.data.add_(p.data, alpha=-group['weight_decay'] * group['lr'])
                p.data.addcdiv_(state['exp_avg'], denom, value=-step_size)
        return loss


def inverse_sqrt_schedule(peak_lr, warmup, floor_lr, power):
    """lr(t) of optim.py:116-133 as one closed form: a straight line from `floor_lr` (t = 0) to `peak_lr` (t = warmup),
    then peak_lr * (warmup / t) ** power.  Returned as a function of the update count."""
    def lr_at(t):
        if t < warmup:
            return floor_lr + (peak_lr - floor_lr) * t / warmup
        return peak_lr * (warmup / t) ** power
    return lr_at


class AdamInverseSqrtWithWarmup(Adam):
    """Adam whose lr follows `inverse_sqrt_schedule` (optim.py:89-139): built at the floor rate, every `step()` counts
    one update per parameter group (`num_updates`, which checkpoints carry) and moves the group's lr along the curve."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, warmup_updates=4000,
                 warmup_init_lr=1e-7, exp_factor=0.5):
        super().__init__(params, lr=warmup_init_lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.schedule = dict(peak_lr=lr, warmup=warmup_updates, floor_lr=warmup_init_lr, power=exp_factor)
        self._lr_at = inverse_sqrt_schedule(**self.schedule)
        for group in self.param_groups:
            group['num_updates'] = 0

    def get_lr_for_step(self, num_updates):
        return self._lr_at(num_updates)

    def step(self, closure=None):
        out = super().step(closure)
        for group in self.param_groups:
            group['num_updates'] += 1
            group['lr'] = self._lr_at(group['num_updates'])
        return out


def clip_grad_norm_(parameters, max_norm, optimizer):
    """Fused stand-in for torch.nn.utils.clip_grad_norm_ at xtrainer.py:225/237 (see
    Adam.clip_grad_norm)."""
    optimizer.clip_grad_norm(max_norm)


# optimizer spec names -> (class, whether the "beta1=..,beta2=.." pair folds into betas=(b1, b2))
_METHODS = {
    'adam': (Adam, True),
    'adam_inverse_sqrt': (AdamInverseSqrtWithWarmup, True),
    'adadelta': (optim.Adadelta, False),
    'adagrad': (optim.Adagrad, False),
    'adamax': (optim.Adamax, False),
    'asgd': (optim.ASGD, False),
    'rmsprop': (optim.RMSprop, False),
    'rprop': (optim.Rprop, False),
    'sgd': (optim.SGD, False),
}
_NUMBER = re.compile(r'^[+-]?(\d+(\.\d*)?|\.\d+)$')


def parse_optimizer_spec(spec):
    """'adam_inverse_sqrt,beta1=0.9,beta2=0.98,lr=0.0001' -> ('adam_inverse_sqrt', {'beta1': 0.9, ...}).
    The string format is the contract of ``--optimizer`` (optim.py:211-229): a method name followed by
    comma-separated ``key=<plain decimal>`` pairs."""
    method, *pairs = spec.split(',')
    kwargs = {}
    for pair in pairs:
        key, eq, val = pair.partition('=')
        if not eq or '=' in val or _NUMBER.match(val) is None:
            raise AssertionError('optimizer option %r is not key=<number>' % pair)
        kwargs[key] = float(val)
    return method, kwargs


def get_optimizer(parameters, s):
    """Drop-in for optim.py:211-270.  adam / adam_inverse_sqrt run on the fused kernels; the torch.optim names the
    reference also accepts pass through (they then see ordinary fp32 parameters).  Unknown methods and options
    the optimizer's constructor does not take raise, like the reference."""
    method, kwargs = parse_optimizer_spec(s)
    if method not in _METHODS:
        raise Exception('Unknown optimization method: "%s"' % method)
    cls, fold_betas = _METHODS[method]
    if fold_betas:
        kwargs['betas'] = (kwargs.pop('beta1', 0.9), kwargs.pop('beta2', 0.999))
    if method == 'sgd':
        assert 'lr' in kwargs
    accepted = list(inspect.signature(cls.__init__).parameters)
    assert accepted[:2] == ['self', 'params']
    unknown = [k for k in kwargs if k not in accepted[2:]]
    if unknown:
        raise Exception('Unexpected parameters: expected "%s", got "%s"' % (str(accepted[2:]), str(list(kwargs))))
    return cls(parameters, **kwargs)
