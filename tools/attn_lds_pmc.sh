#!/bin/bash
# SQ activity of the attention kernels at the benchmarked size (rocprofv3 --pmc on tools/ab_attn.py, one arm, separate passes):
# LDS activity / bank conflicts, issue cycles by instruction class, MFMA pipe busy, waits.   [ARM=libm3p_hip.so]
R=/root/repo
out=$R/gpurun_out/attn_pmc; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
arm=${ARM:-libm3p_hip.so}
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" \
           "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_EXP_GDS"; do
  i=$((i + 1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python $R/tools/ab_attn.py $arm > $out/run$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/root/repo/gpurun_out/attn_pmc/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'attn' in k:
            acc[k.split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    print(k)
    for n, v in sorted(c.items()):
        print('   %-26s %14.0f per launch (n=%d)' % (n, sum(v) / len(v), len(v)))
PY
rm -f $out/p*/*/*kernel_trace.csv
