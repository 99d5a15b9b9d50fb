#!/bin/bash
mkdir -p gpurun_out/r05
bash tools/r04_dp_prof.sh > gpurun_out/r05/dp_prof.txt 2>&1
python - <<'PY' >> gpurun_out/r05/dp_prof.txt
import csv, re
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[r['Name']] = (int(r['Calls']), float(r['TotalDurationNs']) / 1e6)
    return d
a, b = load('gpurun_out/dpprof/plain_kernel_stats.csv'), load('gpurun_out/dpprof/wrapped_kernel_stats.csv')
sa = [c for n, (c, _) in a.items() if 'sumsq' in n][0]
sb = max([c for n, (c, _) in b.items() if 'ce_grad_tile' in n] + [1])
print('steps plain %d wrapped %d; kernel ms per step: plain %.3f wrapped %.3f' % (sa, sb, sum(t for _, t in a.values()) / sa, sum(t for _, t in b.values()) / sb))
rows = []
for n in set(a) | set(b):
    ta = a.get(n, (0, 0))[1] / sa; tb = b.get(n, (0, 0))[1] / sb
    rows.append((tb - ta, n, a.get(n, (0, 0))[0] / sa, b.get(n, (0, 0))[0] / sb, ta, tb))
for d, n, ca, cb, ta, tb in sorted(rows, key=lambda r: -abs(r[0]))[:25]:
    print('%+8.3f ms/step  %-80s launches/step %.1f -> %.1f   %.3f -> %.3f' % (d, n[:80], ca, cb, ta, tb))
PY
