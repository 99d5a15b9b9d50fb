#!/usr/bin/env python3
"""Where the time between kernels goes: reads a rocprofv3 --kernel-trace CSV (one process, one device), takes the
last `--steps` optimizer steps (delimited by the adam kernel), and prints kernel-busy time, wall span and the idle gaps
grouped by (previous kernel -> next kernel).

    python tools/trace_gaps.py gpurun_out/prof/<host>/<pid>_kernel_trace.csv [--steps 3]
"""
import argparse
import csv
import re
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([A-Za-z0-9_:]+(<[^(]*>)?)', name)
    s = m.group(1) if m else name
    return s[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('csv')
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--top', type=int, default=25)
    ap.add_argument('--marker', default='adam_kernel', help='a kernel launched once per step (the sharded data-parallel step launches Adam per shard: use ce_grad_tile_kernel)')
    ap.add_argument('--boundary', default=None, metavar='FROM,TO',
                    help='also list every launch (relative start us, duration us, queue, name) from the last launch of kernel FROM '
                         'to the first launch of kernel TO after it - the step boundary on its streams')
    a = ap.parse_args()
    rows, queue = [], {}
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
            queue[(rows[-1][0], rows[-1][2])] = r.get('Queue_Id', '?')
    rows.sort()
    if a.boundary:
        frm, to = a.boundary.split(',')
        marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
        i0 = max(i for i in range(marks[-2]) if frm in rows[i][2])           # the step before the last full one
        i1 = next(i for i in range(i0 + 1, len(rows)) if to in rows[i][2])
        t0 = rows[i0][0]
        print('step boundary: %s .. %s' % (frm, to))
        for s, e, n in rows[i0:i1 + 1]:
            print('  %9.1f us  %8.1f us  q%-3s %s' % ((s - t0) / 1e3, (e - s) / 1e3, queue[(s, n)], short(n)))
    adam = [i for i, r in enumerate(rows) if a.marker in r[2]]
    assert len(adam) > a.steps, 'not enough steps in the trace'
    lo, hi = adam[-a.steps - 1] + 1, adam[-1] + 1
    seg = rows[lo:hi]
    span = seg[-1][1] - seg[0][0]
    busy = 0
    gaps = defaultdict(lambda: [0, 0])
    end = seg[0][0]
    for s, e, n in seg:
        if s > end:
            pass
        busy += max(0, e - max(s, end))
        end = max(end, e)
    prev_end, prev_name = seg[0][1], seg[0][2]
    for s, e, n in seg[1:]:
        g = s - prev_end
        if g > 0:
            k = (short(prev_name), short(n))
            gaps[k][0] += g
            gaps[k][1] += 1
        if e > prev_end:
            prev_end, prev_name = e, n
    tot_gap = sum(v[0] for v in gaps.values())
    print('steps %d  launches/step %.0f  span/step %.3f ms  busy/step %.3f ms  idle/step %.3f ms'
          % (a.steps, len(seg) / a.steps, span / a.steps / 1e6, busy / a.steps / 1e6, tot_gap / a.steps / 1e6))
    print('%-72s %-72s %8s %6s %8s' % ('previous kernel', 'next kernel', 'us/step', 'n/step', 'avg us'))
    for (p, n), (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:a.top]:
        print('%-72s %-72s %8.1f %6.1f %8.2f' % (p, n, g / a.steps / 1e3, c / a.steps, g / c / 1e3))


if __name__ == '__main__':
    main()
