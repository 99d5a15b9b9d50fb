#!/usr/bin/env python3
"""Which Python lines of the training step still launch torch-native (aten) kernels?

Runs the benchmarked step (bench.build) under torch.profiler with Python stacks and prints, per (aten op, innermost
m3p_amd / bench frame), launches per step and device time - the list `profiles/r02_bench_kernel_stats.csv` shows as ~100
`at::native` / rocprim / copyBuffer launches per step.

    python tools/native_ops.py [--batch 256] [--steps 3] > gpurun_out/native_ops.txt
"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--steps', type=int, default=3)
    args = ap.parse_args()
    torch.set_num_threads(4)
    import bench
    from m3p_amd import synth
    cfg = dict(synth.CONFIGS['cfg2'])
    cfg['B'] = args.batch
    trainer, tup = bench.build(cfg, 0.1, 1, 0, 0)

    def step():
        trainer.pretrain_under_step(tup, 'google', 't2i', 'en', 1.0, 1.0, 1.0, 1.0)
        trainer.n_iter += 1
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    # (1) every aten op the step's own thread dispatches that touches device memory, by calling line
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    VIEWS = ('view', 'reshape', 'transpose', 'select', 'slice', 'as_strided', 'empty', 'detach', 'alias', 't.default',
             'expand', 'unsqueeze', 'squeeze', 'permute', '_unsafe_view', 'narrow', 'unbind', 'split', 'chunk')
    agg = collections.defaultdict(int)

    class Log(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            name = str(func)
            outs = out if isinstance(out, (tuple, list)) else (out,)
            if any(torch.is_tensor(o) and o.is_cuda for o in outs) and not any(v in name for v in VIEWS):
                where = '?'
                for fr in reversed(traceback.extract_stack()):
                    if ('m3p_amd' in fr.filename or fr.filename.endswith('bench.py')) and 'native_ops' not in fr.filename:
                        where = '%s:%d %s' % (fr.filename.replace(ROOT + '/', ''), fr.lineno, fr.name)
                        break
                agg[(name, where)] += 1
            return out
    with Log():
        for _ in range(args.steps):
            step()
    torch.cuda.synchronize()
    print('aten ops dispatched on the stepping thread (backward runs on the autograd thread and is not in this list):')
    for (name, where), n in sorted(agg.items(), key=lambda kv: -kv[1]):
        print('%7.2f /step  %-40s %s' % (n / args.steps, name[:40], where))
    # (2) all threads: launches per step by kernel family
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
    fam = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA:
            k = ev.name
            key = 'torch/' + k.split('at::native::')[1][:60] if 'at::native::' in k else ('runtime/' + k[:40] if k.startswith('__amd') or 'rocprim' in k or 'Memcpy' in k or 'Memset' in k else 'm3p')
            fam[key][0] += 1
            fam[key][1] += ev.device_time if hasattr(ev, 'device_time') else 0.0
    tot = sum(v[0] for k, v in fam.items() if k != 'm3p') / args.steps
    print('\nnon-m3p device launches per step: %.1f   (m3p kernels: %.1f)' % (tot, fam['m3p'][0] / args.steps))
    for k, (n, t) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        if k != 'm3p':
            print('%7.2f /step %8.1f us  %s' % (n / args.steps, t / args.steps, k))


if __name__ == '__main__':
    main()
