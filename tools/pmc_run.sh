#!/bin/bash
# usage: tools/pmc_run.sh <outdir-under-gpurun_out> <kernel-name-substring> -- <cmd...>
# Runs the command once per counter group (separate rocprofv3 --pmc passes, kernel-trace only)
# and prints per-kernel averages for kernels whose name contains the substring.
out=$1; pat=$2; shift 3
cd /tmp; export TMPDIR=/tmp
groups=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN")
i=0
for g in "${groups[@]}"; do
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d /root/repo/gpurun_out/$out/g$i -- "$@" > /dev/null 2>&1
  f=$(ls /root/repo/gpurun_out/$out/g$i/*/*counter_collection.csv 2>/dev/null | head -1)
  python3 - "$f" "$pat" <<'PY'
import csv, sys, collections
f, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
except Exception as e:
    print('no data', f, e)
for k, v in acc.items():
    print('%-28s n=%d mean %.4g' % (k, len(v), sum(v) / len(v)))
PY
  i=$((i+1))
done
