#!/bin/bash
# MFMA utilisation, wait share and sustained clock of the vendor's kernel and ours on the same shapes
R=$(cd "$(dirname "$0")/../.." && pwd)
out=$R/gpurun_out/pmc; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for shape in "768 3072" "3072 768"; do
  for which in vendor 1 2; do
    tag=$(echo "$which $shape" | tr ' ' '_')
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $out/sq_$tag -- python $R/tools/gemm_pmc.py $which $shape > /dev/null 2>&1
    rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/clk_$tag -- python $R/tools/gemm_pmc.py $which $shape > /dev/null 2>&1
    rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $out/x_$tag -- python $R/tools/gemm_pmc.py $which $shape > /dev/null 2>&1
  done
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get('OUT', '/root/repo/gpurun_out/pmc')
def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, d, '*', '*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
            dur[r['Kernel_Name']].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
    return acc, dur
lines = []
for d in sorted(os.listdir(out)):
    if not d.startswith('sq_'):
        continue
    tag = d[3:]
    sq, dur = load(d)
    ck, cdur = load('clk_' + tag)
    xx, _ = load('x_' + tag)
    for k in sq:
        if 'gemm' not in k and 'Cijk' not in k:
            continue
        s = sq[k]
        n = len(s['SQ_BUSY_CU_CYCLES'])
        util = sum(s['SQ_VALU_MFMA_BUSY_CYCLES']) / (4 * sum(s['SQ_BUSY_CU_CYCLES']))
        wait = sum(s['SQ_WAIT_ANY']) / sum(s['SQ_WAVE_CYCLES'])
        g = ck.get(k, {}).get('GRBM_GUI_ACTIVE', [])
        ns = cdur.get(k, [])
        ghz = sum(g) / 8.0 / (sum(ns) + len(ns) * 7000.0) if g else float('nan')
        extra = ' '.join('%s=%.3g' % (c, sum(v) / len(v)) for c, v in sorted(xx.get(k, {}).items()))
        lines.append('%-14s %-60s n=%d  %.1f us  MFMA util %.1f%%  wait %.0f%%  clock %.2f GHz  %s' % (tag, k[:60], n, sum(dur[k]) / len(dur[k]) / 1e3, 100 * util, 100 * wait, ghz, extra))
open(os.path.join(out, 'summary.txt'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
find $out -name "*.csv" -size +200k -delete; find $out -name "*.db" -delete
