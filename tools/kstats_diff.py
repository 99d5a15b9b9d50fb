#!/usr/bin/env python3
"""Per-kernel average duration of two rocprofv3 --stats runs side by side (A/B of two library builds).
usage: python tools/kstats_diff.py A_kernel_stats.csv B_kernel_stats.csv"""
import csv
import re
import sys


def load(path):
    out = {}
    for r in csv.DictReader(open(path)):
        name = re.sub(r'\(anonymous namespace\)::', '', r['Name'])
        name = re.sub(r'^void ', '', name)
        name = re.sub(r'\(.*$', '', name)[:60]
        out[name] = (float(r['AverageNs']) / 1e3, int(r['Calls']), float(r['TotalDurationNs']) / 1e6)
    return out


a, b = load(sys.argv[1]), load(sys.argv[2])
print('%-62s %9s %9s %7s %9s' % ('kernel', 'A us', 'B us', 'B/A', 'dTotal ms'))
tot = 0.0
for k in sorted(a, key=lambda k: -a[k][2]):
    if k in b and a[k][2] > 0.05:
        d = b[k][2] - a[k][2]
        tot += d
        print('%-62s %9.1f %9.1f %7.3f %9.2f' % (k, a[k][0], b[k][0], b[k][0] / a[k][0], d))
print('sum of differences over the run: %.2f ms' % tot)
