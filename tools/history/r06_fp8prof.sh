#!/bin/bash
# kernel census of the cfg4 step, bf16 against --fp8 (where the 8-bit path's time goes)
R=$(cd "$(dirname "$0")/../.." && pwd)
out=$R/gpurun_out/fp8prof; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bf16 -- python $R/bench.py --config cfg4 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline > $out/bf16.log 2>&1
M3P_FP8_SITES=${FP8_SITES:-} rocprofv3 --kernel-trace --stats --output-format csv -d $out/fp8 -- python $R/bench.py --config cfg4 --batch 64 --fp8 --steps 10 --warmup 3 --no-cpu-baseline > $out/fp8.log 2>&1
tail -1 $out/bf16.log | cut -c1-200; tail -1 $out/fp8.log | cut -c1-200
cd $R
cp $(ls $out/bf16/*/*kernel_stats.csv | head -1) $out/bf16_kernel_stats.csv; cp $(ls $out/fp8/*/*kernel_stats.csv | head -1) $out/fp8_kernel_stats.csv
find $out -name "*kernel_trace.csv" -delete
python - <<'PY' > $out/diff.txt
import csv
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[r['Name']] = (int(r['Calls']), float(r['TotalDurationNs']) / 1e6)
    return d
a, b = load('gpurun_out/fp8prof/bf16_kernel_stats.csv'), load('gpurun_out/fp8prof/fp8_kernel_stats.csv')
sa = [c for n, (c, _) in a.items() if 'sumsq' in n][0]; sb = [c for n, (c, _) in b.items() if 'sumsq' in n][0]
print('steps bf16 %d fp8 %d; kernel ms per step: bf16 %.3f fp8 %.3f' % (sa, sb, sum(t for _, t in a.values()) / sa, sum(t for _, t in b.values()) / sb))
rows = []
for n in set(a) | set(b):
    ta = a.get(n, (0, 0))[1] / sa; tb = b.get(n, (0, 0))[1] / sb
    rows.append((tb - ta, n, a.get(n, (0, 0))[0] / sa, b.get(n, (0, 0))[0] / sb, ta, tb))
for d, n, ca, cb, ta, tb in sorted(rows, key=lambda r: -abs(r[0]))[:40]:
    print('%+8.3f ms/step  %-90s launches/step %.1f -> %.1f   %.3f -> %.3f' % (d, n[:90], ca, cb, ta, tb))
PY
cat $out/diff.txt
