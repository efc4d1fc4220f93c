#!/bin/bash
# PMC passes over tools/inr_bench.py (one process per counter group); prints per-kernel means.
# usage: bash tools/pmc_inr.sh <tag> [n]
TAG=$1; N=${2:-16777216}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_inr_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16" \
         "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/set$i -- python3 $REPO/tools/inr_bench.py $N > $OUT/set$i.log 2>&1 || echo "set $i failed" >> $OUT/errors.log
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + '/set*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'inr_' in r['Kernel_Name'] and 'pack' not in r['Kernel_Name']:
            agg[(r['Kernel_Name'][:60], r['Counter_Name'])].append(float(r['Counter_Value']))
with open(out + '/summary.txt', 'w') as fh:
    for k in sorted(agg):
        line = f"{k[0]:62s} {k[1]:34s} n={len(agg[k])} mean={sum(agg[k])/len(agg[k]):.5g}"
        print(line); fh.write(line + "\n")
PY
