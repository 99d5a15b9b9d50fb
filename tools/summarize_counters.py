#!/usr/bin/env python3
"""Turns gpurun_out/counters/ (tools/collect_counters.sh) into profiles/<ROUND>_traffic.json and
profiles/<ROUND>_counters.md (ROUND defaults to r02): per kernel, launches per step, average duration, HBM bytes per launch
(2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md, calibrated on the GELU
pass), achieved HBM GB/s, and MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, 'gpurun_out', os.environ.get('COUNTERS_DIR', 'counters'))

def newest(pattern):
    """gpurun merges a call's files into what earlier calls left behind: only the newest run of a pass counts"""
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:]

def counters(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in newest(os.path.join(src, sub, '*', '*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    return acc

stats = {}
for f in newest(os.path.join(src, 'stats', '*', '*kernel_stats.csv')):
    for r in csv.DictReader(open(f)):
        stats[r['Name']] = (int(r['Calls']), float(r['AverageNs']))
fetch, write, sq, tcc = counters('fetch'), counters('write'), counters('sq'), counters('tcc')

# sustained clock per kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs by rocprofv3) / the dispatch's own duration in the SAME
# pass.  The counter also runs over the dispatch's set-up and tear-down - about OVH_US microseconds per dispatch in a counter
# pass (calibrated on the memory-bound kernels, which run at the 2.4 GHz ceiling: adam 1650 us reads 2.42 GHz raw, the 16-us
# weight-gradient reduction 3.47) - so short kernels read high; the column is corrected by it and kernels under 40 us are
# not reported.
OVH_US = 7.0
clk = collections.defaultdict(lambda: [0.0, 0.0, 0])
for f in newest(os.path.join(src, 'clk', '*', '*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            e = clk[r['Kernel_Name']]
            e[0] += float(r['Counter_Value']); e[1] += float(r['End_Timestamp']) - float(r['Start_Timestamp']); e[2] += 1
def clock_ghz(name):
    c, ns, n = clk.get(name, (0, 0, 0))
    if n == 0 or ns / n < 40e3:
        return None
    return c / 8.0 / (ns + n * OVH_US * 1e3)
TAG = os.environ.get('ROUND', 'r04')
# optimizer steps in the profiled run: the global-norm kernel runs once per step (the fused Adam pass is two launches a step since
# round 5 - the lazily zeroed vocabulary range is its own piece)
steps = float(max([c for n, (c, _) in stats.items() if 'sumsq' in n] + [1]))
rows, kernels = [], {}
for name, (calls, avg_ns) in sorted(stats.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    fk = fetch.get(name, {}).get('FETCH_SIZE')
    wk = write.get(name, {}).get('WRITE_SIZE')
    if not fk or not wk or calls * avg_ns < 2e5:
        continue
    f_kb, w_kb = sum(fk) / len(fk), sum(wk) / len(wk)
    hbm = (2 * f_kb + w_kb) * 1024
    s = sq.get(name, {})
    mf = s.get('SQ_VALU_MFMA_BUSY_CYCLES'); bc = s.get('SQ_BUSY_CU_CYCLES')
    util = (sum(mf) / len(mf)) / (4.0 * sum(bc) / len(bc)) if mf and bc and sum(bc) > 0 else None
    wc = s.get('SQ_WAVE_CYCLES'); wa = s.get('SQ_WAIT_ANY')
    wait = (sum(wa) / sum(wc)) if wc and wa and sum(wc) > 0 else None
    t = tcc.get(name, {})
    th, tm_ = t.get('TCC_HIT_sum'), t.get('TCC_MISS_sum')
    l2hit = sum(th) / (sum(th) + sum(tm_)) if th and tm_ and sum(th) + sum(tm_) > 0 else None
    ghz = clock_ghz(name)
    kernels[name[:90]] = dict(launches=calls, fetch_kb=round(f_kb, 1), write_kb=round(w_kb, 1), hbm_bytes_per_launch=int(hbm),
                              clock_ghz=None if ghz is None else round(ghz, 3), l2_hit=None if l2hit is None else round(l2hit, 3))
    rows.append((name, calls / steps, avg_ns / 1e3, hbm, hbm / avg_ns, util, wait, ghz, l2hit))

# the dominant instance of bench.py's roofline block: the dGELU data gradient (epilogue 5 runs only on this shape in the
# step), on whichever NT kernel the dispatch picked
dg = [k for k in kernels if 'gemm_nt_w8_kernel<5' in k and 'true' not in k.split('gemm_nt_w8_kernel<5')[1][:8]] or [k for k in kernels if 'gemm_nt_ring_kernel<5>' in k]
bench_keys = {'gemm_nt/dgelu M=41984 N=3072 K=768': kernels[dg[0]]['hbm_bytes_per_launch']} if dg else {}
# round 4: the data gradient through gelu' reads one-byte codes (epilogue 7 runs only on this shape in the step)
mq = [k for k in kernels if 'gemm_nt_w8_kernel<7' in k]
if mq:
    bench_keys['gemm_nt/mulq M=41984 N=3072 K=768'] = kernels[mq[0]]['hbm_bytes_per_launch']
# ... and lin1 + GELU + byte (epilogue 8) is the other instance that only this shape runs: the bench line's dominant kernel
gq = [k for k in kernels if 'gemm_nt_w8_kernel<8' in k]
if gq:
    bench_keys['gemm_nt/bias_geluq M=41984 N=3072 K=768'] = kernels[gq[0]]['hbm_bytes_per_launch']
# the weight-gradient kernel runs several shapes under one name: no per-shape traffic (bench.py prints null for it)
# sustained clock over the GEMM kernels, launch-time weighted: the secondary roofline of bench.py
gw = [(kernels[k]['clock_ghz'], stats_calls * stats_ns) for k, (stats_calls, stats_ns) in
      ((n[:90], stats[n]) for n in stats if n[:90] in kernels) if 'gemm_' in k and kernels[k]['clock_ghz']]
gemm_clock = sum(c * w for c, w in gw) / sum(w for _, w in gw) if gw else None
note = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tools/collect_counters.sh) of `python bench.py --steps 3 "
        "--warmup 1 --no-cpu-baseline`; values are KB per launch averaged over all launches of the kernel. hbm_bytes = (2*FETCH_SIZE + "
        "WRITE_SIZE)*1024: the x2 on FETCH_SIZE is the gfx950 correction of MI355X_MICROARCH.md (HBM section), calibrated here on "
        "gelu_fwd_kernel (streams 258 MB in, 258 MB out).")
json.dump(dict(_note=note, kernels=kernels, bench_keys=bench_keys, gemm_clock_ghz=None if gemm_clock is None else round(gemm_clock, 3)), open(os.path.join(ROOT, 'profiles', TAG + '_traffic.json'), 'w'), indent=1)
with open(os.path.join(ROOT, 'profiles', TAG + '_counters.md'), 'w') as o:
    o.write('# ' + TAG + ' - per-kernel counters of one cfg2 training step (MI355X, rocprofv3, separate --pmc passes)\n\n')
    o.write('Collected by `tools/collect_counters.sh`, summarised by `tools/summarize_counters.py`.  Durations come from the plain\n'
            '`--kernel-trace --stats` pass (counter passes serialise kernels).  HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950\n'
            'correction, MI355X_MICROARCH.md); HBM GB/s = bytes / duration against the 8 TB/s peak (about 6.3 TB/s achievable).\n'
            'MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES): the share of SIMD-cycles, at the clock the\n'
            'kernel actually ran at, in which the matrix pipe was busy.  wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES.\n'
            'clock = GRBM_GUI_ACTIVE / 8 XCDs / (dispatch duration + %.0f us of dispatch set-up the counter also covers), own pass;\n'
            'the ceiling is 2.4 GHz, and clock x 256 CUs x 4096 FLOP/clk is the matrix peak at the clock the kernel really ran at\n'
            '(launch-time-weighted over the GEMM kernels: %s GHz -> %s TFLOP/s; the 2.5 PF of the roofline assumes 2.4 GHz).\n'
            'L2 hit = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum); the misses x 128 B reproduce the HBM column, i.e. FETCH/WRITE_SIZE\n'
            'count what leaves the XCDs L2s - Infinity-Cache hits included; no counter on this stack separates those from HBM.\n\n'
            % (OVH_US, '%.2f' % gemm_clock if gemm_clock else '-', '%.0f' % (gemm_clock * 256 * 4096 / 1e3) if gemm_clock else '-'))
    o.write('| kernel | launches / step | avg us | HBM MB / launch | HBM GB/s | MFMA util | wait | clock GHz | L2 hit |\n|---|---|---|---|---|---|---|---|---|\n')
    for name, lps, us, hbm, gbs, util, wait, ghz, l2hit in rows:
        short = name.replace('(anonymous namespace)::', '').replace('void ', '')
        short = short.split('(')[0][:60]
        o.write('| `%s` | %.1f | %.1f | %.1f | %.0f | %s | %s | %s | %s |\n' % (short, lps, us, hbm / 1e6, gbs, '%.1f %%' % (100 * util) if util is not None else '-',
                                                                                '%.0f %%' % (100 * wait) if wait is not None else '-',
                                                                                '%.2f' % ghz if ghz is not None else '-',
                                                                                '%.0f %%' % (100 * l2hit) if l2hit is not None else '-'))
print('wrote profiles/%s_traffic.json, profiles/%s_counters.md;' % (TAG, TAG), len(rows), 'kernels')
