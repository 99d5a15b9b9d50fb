"""Small host-side helpers of the trainer surface (M3P/src/utils.py): device transfer, the
scheduled lambda coefficients of ``train_x.py`` and boolean flags.  Pure Python, no kernels."""
import os

import torch

# coefficients that may carry a schedule "it0:v0,it1:v1,..." (utils.py:28-30)
DYNAMIC_COEFF = ['lambda_mlm', 'lambda_mass', 'lambda_ic', 'lambda_imlm', 'lambda_ida', 'lambda_tifg', 'lambda_rel',
                 'lambda_mrm', 'lambda_mrfr', 'lambda_t2i', 'lambda_i2t']


_PIN_H2D = os.environ.get('M3P_PIN_H2D', '1') != '0'      # developer switch for A/B runs


def _h2d(x):
    """Host tensor -> current device through page-locked memory: a copy from pageable memory blocks the host until every
    kernel queued before it has run (one full pipeline drain per training step: the GPU then idles ~0.2 ms while the
    host starts enqueueing the next step); from pinned memory (PyTorch's caching host allocator) it is just another
    stream-ordered operation."""
    if x.is_cuda:
        return x
    if _PIN_H2D and torch.cuda.is_available() and not x.is_pinned() and x.numel() > 0:
        x = x.pin_memory()
    return x.cuda(non_blocking=True)


def to_cuda(*args):
    """utils.py:233-237: None stays None, tensors move to the current device without blocking."""
    return [None if x is None else _h2d(x) for x in args]


def _parse_schedule(spec):
    """'3' -> (3.0, None);  '0:1,1000:0' -> (1.0, [(0, 1.0), (1000, 0.0)])  (utils.py:249-268)."""
    if isinstance(spec, (int, float)):
        return float(spec), None
    knots = spec.split(',')
    if len(knots) == 1:
        return float(spec), None
    pts = []
    for knot in knots:
        it, _, val = knot.partition(':')
        assert it.isdigit() and val != '', 'bad lambda schedule %r' % spec
        pts.append((int(it), float(val)))
    assert all(a[0] < b[0] for a, b in zip(pts, pts[1:])), 'lambda schedule iterations must increase: %r' % spec
    return pts[0][1], pts


def parse_lambda_config(params):
    """Turns every ``params.lambda_*`` string into its initial float and stores the schedule
    (or None) as ``params.lambda_*_config``.  Coefficients a caller did not define are skipped."""
    for name in DYNAMIC_COEFF:
        if not hasattr(params, name):
            continue
        value, config = _parse_schedule(getattr(params, name))
        setattr(params, name, value)
        setattr(params, name + '_config', config)


def get_lambda_value(config, n_iter):
    """Piecewise-linear interpolation of a schedule at iteration n_iter; constant after the
    last knot (utils.py:271-283)."""
    if n_iter >= config[-1][0]:
        return config[-1][1]
    for (x_a, y_a), (x_b, y_b) in zip(config, config[1:]):
        if x_a <= n_iter < x_b:
            return y_a + (n_iter - x_a) * float(y_b - y_a) / float(x_b - x_a)
    raise AssertionError('iteration %d precedes the schedule %r' % (n_iter, config))


def update_lambdas(params, n_iter):
    """utils.py:286-293."""
    for name in DYNAMIC_COEFF:
        config = getattr(params, name + '_config', None)
        if config is not None:
            setattr(params, name, get_lambda_value(config, n_iter))


def concat_rows(tensors):
    return torch.cat([t.reshape(-1) for t in tensors])


def concat_batches(x1, len1, lang1_id, x2, len2, lang2_id, pad_idx, eos_idx, reset_positions):
    """utils.py:324-349: the two sentences of every column joined into one sequence (TLM input).  With
    ``reset_positions`` the second sentence follows the first's closing delimiter and its positions restart at 0;
    without (the same-language denoising case) it overwrites that delimiter and positions run on.
    -> (x (slen, bs), lengths, positions (slen, bs), langs (slen, bs))."""
    assert not reset_positions or lang1_id != lang2_id
    lengths = len1 + len2 - (0 if reset_positions else 1)
    slen, bs = int(lengths.max()), lengths.size(0)
    start = len1 if reset_positions else len1 - 1            # where sentence 2 begins, per column
    t = torch.arange(slen, device=x1.device)[:, None]        # (slen, 1)
    second = t >= start[None, :]                             # (slen, bs): rows that belong to sentence 2 (or the padding behind it)
    x = x1.new_full((slen, bs), pad_idx)
    x[:x1.size(0)] = x1
    src = (t - start[None, :]).clamp_(0, x2.size(0) - 1)
    from2 = second & (t < (start + len2)[None, :])
    x = torch.where(from2, torch.gather(x2, 0, src), x)
    positions = t.repeat(1, bs)
    if reset_positions:
        positions = torch.where(second, positions - len1[None, :], positions)
    langs = torch.where(second, torch.full_like(x, lang2_id), torch.full_like(x, lang1_id))
    assert int((x == eos_idx).sum()) == (4 if reset_positions else 3) * bs
    return x, lengths, positions, langs
