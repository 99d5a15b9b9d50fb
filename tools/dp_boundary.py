#!/usr/bin/env python3
"""What the HOST does while the GPU idles at the step boundary of the wrapped data-parallel step: reads a rocprofv3
--kernel-trace --hip-runtime-trace pair of CSVs, takes the window from the end of a mid-run Adam launch to the next
launch of `--to`, and lists the kernels (all queues) and the HIP runtime calls that overlap it.

    python tools/dp_boundary.py <dir with *_kernel_trace.csv and *_hip_api_trace.csv> [--to seq_masks_kernel]
"""
import argparse
import csv
import glob
import re


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([A-Za-z0-9_:]+(<[^(]*>)?)', name)
    return (m.group(1) if m else name)[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dir')
    ap.add_argument('--frm', default='adam_ranges_kernel')
    ap.add_argument('--to', default='seq_masks_kernel')
    ap.add_argument('--min-us', type=float, default=4.0, help='list runtime calls at least this long (all are counted)')
    a = ap.parse_args()
    kt = glob.glob(a.dir + '/**/*kernel_trace.csv', recursive=True)[0]
    ht = glob.glob(a.dir + '/**/*hip_api_trace.csv', recursive=True)[0]
    ker = []
    for r in csv.DictReader(open(kt)):
        ker.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')))
    ker.sort()
    frm = [i for i, k in enumerate(ker) if a.frm in k[2]]
    i0 = frm[len(frm) // 2]                                  # a step in the middle of the run (the timed region)
    i1 = next(i for i in range(i0 + 1, len(ker)) if a.to in ker[i][2])
    t0, t1 = ker[i0][1], ker[i1][0]
    print('window: end of %s .. start of %s = %.1f us' % (a.frm, a.to, (t1 - t0) / 1e3))
    for s, e, n, q in ker[i0:i1 + 1]:
        print('  kernel %9.1f us  %8.1f us  q%-3s %s' % ((s - t0) / 1e3, (e - s) / 1e3, q, short(n)))
    calls = []
    for r in csv.DictReader(open(ht)):
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if e >= t0 - 200000 and s <= t1:
            calls.append((s, e, r['Function'], r['Thread_Id']))
    calls.sort()
    print('runtime calls overlapping [window - 200 us, window end]: %d' % len(calls))
    tot = {}
    for s, e, f, t in calls:
        d = tot.setdefault((t, f), [0, 0])
        d[0] += 1
        d[1] += e - s
        if (e - s) / 1e3 >= a.min_us:
            print('  call   %9.1f us  %8.1f us  t%-8s %s' % ((s - t0) / 1e3, (e - s) / 1e3, t[-5:], f))
    print('by (thread, call):')
    for (t, f), (c, ns) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print('  t%-8s %-40s n %4d  %9.1f us' % (t[-5:], f, c, ns / 1e3))


if __name__ == '__main__':
    main()
