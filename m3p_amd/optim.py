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

from . import ops


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
            if cur is not None and cur['end'] == o and cur['step'] == step and cur['group'] is group:
                cur['end'] = end
                cur['params'].append(p)
            else:
                cur = dict(start=o, end=end, step=step, group=group, params=[p])
                ranges.append(cur)
        return ranges

    def clip_grad_norm(self, max_norm):
        """clip_grad_norm_(parameters, max_norm) of xtrainer.py:225, deferred: the global
        sum of squares is reduced on the device now, the scaling is applied inside the Adam
        kernel (no host sync, no extra pass over the gradients)."""
        for ent in self._arenas.values():
            arena = ent['arena']
            if arena.model.ddp_hook is not None:
                arena.model.ddp_hook.finish()
            ent['gnorm'].zero_()
            for r in self._active_ranges(arena):
                ops.sumsq(arena.grad[r['start']:r['end']], ent['gnorm'])
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
            for r in self._active_ranges(arena):
                group = r['group']
                beta1, beta2 = group['betas']
                step = r['step'] + 1
                bc1 = 1 - beta1 ** step
                bc2 = 1 - beta2 ** step
                step_size = group['lr'] * math.sqrt(bc2) / bc1
                s, e = r['start'], r['end']
                ops.adam_step(arena.master[s:e], arena.grad[s:e], ent['m'][s:e], ent['v'][s:e], arena.w16[s:e],
                              group['lr'], beta1, beta2, group['eps'], group['weight_decay'], step_size,
                              gnorm_sq=ent['gnorm'] if max_norm > 0 else None, max_norm=max_norm,
                              grad_scale=self.grad_scale, zero_grad=True)
                for p in r['params']:
                    self.state[p]['step'] = step
            # untouched ranges may still hold stale values only if someone wrote them by hand
            arena.after_fused_step()
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


class AdamInverseSqrtWithWarmup(Adam):
    """optim.py:89-139."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, warmup_updates=4000,
                 warmup_init_lr=1e-7, exp_factor=0.5):
        super().__init__(params, lr=warmup_init_lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.warmup_updates = warmup_updates
        self.warmup_init_lr = warmup_init_lr
        warmup_end_lr = lr
        self.lr_step = (warmup_end_lr - warmup_init_lr) / warmup_updates
        self.exp_factor = exp_factor
        self.decay_factor = warmup_end_lr * warmup_updates ** self.exp_factor
        for param_group in self.param_groups:
            param_group['num_updates'] = 0

    def get_lr_for_step(self, num_updates):
        if num_updates < self.warmup_updates:
            return self.warmup_init_lr + num_updates * self.lr_step
        return self.decay_factor * (num_updates ** -self.exp_factor)

    def step(self, closure=None):
        super().step(closure)
        for param_group in self.param_groups:
            param_group['num_updates'] += 1
            param_group['lr'] = self.get_lr_for_step(param_group['num_updates'])


def clip_grad_norm_(parameters, max_norm, optimizer):
    """Fused stand-in for torch.nn.utils.clip_grad_norm_ at xtrainer.py:225/237 (see
    Adam.clip_grad_norm)."""
    optimizer.clip_grad_norm(max_norm)


def get_optimizer(parameters, s):
    """optim.py:211-270: "adam_inverse_sqrt,beta1=0.9,beta2=0.98,lr=0.0001" style specs.
    adam / adam_inverse_sqrt run on the fused kernels; torch.optim pass-throughs are kept
    for completeness (they then see ordinary fp32 parameters)."""
    if ',' in s:
        method = s[:s.find(',')]
        optim_params = {}
        for x in s[s.find(',') + 1:].split(','):
            split = x.split('=')
            assert len(split) == 2
            assert re.match(r'^[+-]?(\d+(\.\d*)?|\.\d+)$', split[1]) is not None
            optim_params[split[0]] = float(split[1])
    else:
        method = s
        optim_params = {}
    if method in ('adam', 'adam_inverse_sqrt'):
        optim_fn = Adam if method == 'adam' else AdamInverseSqrtWithWarmup
        optim_params['betas'] = (optim_params.get('beta1', 0.9), optim_params.get('beta2', 0.999))
        optim_params.pop('beta1', None)
        optim_params.pop('beta2', None)
    elif method == 'adadelta':
        optim_fn = optim.Adadelta
    elif method == 'adagrad':
        optim_fn = optim.Adagrad
    elif method == 'adamax':
        optim_fn = optim.Adamax
    elif method == 'asgd':
        optim_fn = optim.ASGD
    elif method == 'rmsprop':
        optim_fn = optim.RMSprop
    elif method == 'rprop':
        optim_fn = optim.Rprop
    elif method == 'sgd':
        optim_fn = optim.SGD
        assert 'lr' in optim_params
    else:
        raise Exception('Unknown optimization method: "%s"' % method)
    expected_args = list(inspect.signature(optim_fn.__init__).parameters.keys())
    assert expected_args[:2] == ['self', 'params']
    if not all(k in expected_args[2:] for k in optim_params.keys()):
        raise Exception('Unexpected parameters: expected "%s", got "%s"' % (str(expected_args[2:]), str(optim_params.keys())))
    return optim_fn(parameters, **optim_params)
