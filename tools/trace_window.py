#!/usr/bin/env python3
"""Kernels (and memory copies) around the largest idle gaps of one step of a rocprofv3 trace: name, start offset, duration,
gap to the previous end.   python tools/trace_window.py <kernel_trace.csv> [memory_copy_trace.csv] [--marker K] [--min-gap us]"""
import csv, re, sys
args = [a for a in sys.argv[1:] if not a.startswith('--')]
marker = 'ce_grad_tile_kernel'
min_gap = 30.0
frm = to = None
for i, a in enumerate(sys.argv):
    if a == '--marker': marker = sys.argv[i + 1]
    if a == '--from': frm = sys.argv[i + 1]
    if a == '--to': to = sys.argv[i + 1]
    if a == '--min-gap': min_gap = float(sys.argv[i + 1])
rows = []
for r in csv.DictReader(open(args[0])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'\(anonymous namespace\)::|^void ', '', r['Kernel_Name'])[:60]))
if len(args) > 1:
    for r in csv.DictReader(open(args[1])):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY %s %s B' % (r.get('Direction', ''), r.get('Bytes', r.get('Size', '?')))))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
lo, hi = marks[-2], marks[-1]
t0 = rows[lo][0]
end = rows[lo][1]
if frm:      # everything between the first FROM kernel of the step and the next TO kernel, with the gap in front of each
    a = next(i for i in range(lo, len(rows)) if frm in rows[i][2])
    b = next(i for i in range(a, len(rows)) if to in rows[i][2])
    end = rows[a - 1][1]
    for s_, e_, n_ in rows[a:b + 2]:
        print('%9.1f us  dur %7.1f  gap %6.1f  %s' % ((s_ - t0) / 1e3, (e_ - s_) / 1e3, (s_ - end) / 1e3, n_))
        end = max(end, e_)
    sys.exit(0)
for i in range(lo + 1, hi + 1):
    s, e, n = rows[i]
    gap = (s - end) / 1e3
    if gap > min_gap:
        for j in range(max(lo, i - 3), min(hi, i + 2) + 1):
            sj, ej, nj = rows[j]
            print('%9.1f us  %8.1f us  %s%s' % ((sj - t0) / 1e3, (ej - sj) / 1e3, nj, '   <-- after a %.0f us gap' % gap if j == i else ''))
        print('   ...')
    end = max(end, e)
