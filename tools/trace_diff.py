#!/usr/bin/env python3
"""Per-kernel time of two rocprofv3 --kernel-trace CSVs side by side (A = wrapped, B = plain), per step over the last
`--steps` steps (delimited by `--marker`): which kernels exist only in A, and which ran longer there.

    python tools/trace_diff.py A_kernel_trace.csv B_kernel_trace.csv [--steps 3]
"""
import argparse
import csv
import re
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([A-Za-z0-9_:]+(<[^(]*>)?)', name)
    return (m.group(1) if m else name)[:64]


def per_step(path, steps, marker):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    seg = rows[marks[-steps - 1] + 1:marks[-1] + 1]
    tot = defaultdict(lambda: [0, 0.0])
    for s, e, n in seg:
        d = tot[short(n)]
        d[0] += 1
        d[1] += (e - s) / 1e3
    return {k: (c / steps, us / steps) for k, (c, us) in tot.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('a')
    ap.add_argument('b')
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--top', type=int, default=20)
    ap.add_argument('--marker', default='ce_grad_tile_kernel')
    a = ap.parse_args()
    A, B = per_step(a.a, a.steps, a.marker), per_step(a.b, a.steps, a.marker)
    rows = [(A.get(k, (0, 0))[1] - B.get(k, (0, 0))[1], k) for k in set(A) | set(B)]
    print('sum of kernel durations per step: A %.1f us  B %.1f us  A - B %.1f us' %
          (sum(v[1] for v in A.values()), sum(v[1] for v in B.values()), sum(r[0] for r in rows)))
    print('%-66s %8s %8s %8s %8s %9s' % ('kernel', 'n A', 'us A', 'n B', 'us B', 'A-B us'))
    for d, k in sorted(rows, key=lambda r: -abs(r[0]))[:a.top]:
        print('%-66s %8.1f %8.1f %8.1f %8.1f %9.1f' % (k, *A.get(k, (0, 0)), *B.get(k, (0, 0)), d))


if __name__ == '__main__':
    main()
