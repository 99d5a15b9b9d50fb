import torch
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for mb in (64, 258, 1024, 4096):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device='cuda'); b = torch.empty_like(a)
    ms = t(lambda: a.zero_()); print('fill  %5d MB: %.3f ms  %.2f TB/s (write)' % (mb, ms, mb / 1024 / 1024 * 1.048576 / ms * 1e3 / 1e3 * 1000 / 1000 * 1e0 if False else mb * 1.048576e6 / ms / 1e9))
    ms = t(lambda: b.copy_(a)); print('copy  %5d MB: %.3f ms  %.2f TB/s (read+write)' % (mb, ms, 2 * mb * 1.048576e6 / ms / 1e9))
    ms = t(lambda: a.sum()); print('sum   %5d MB: %.3f ms  %.2f TB/s (read)' % (mb, ms, mb * 1.048576e6 / ms / 1e9))
