#!/bin/bash
# rocprofv3 kernel trace + MFMA counters for the INR forward kernel.  Usage: bash tools/profile_inr.sh <tag>
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_inr_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/inr_bench.py > $OUT/bench.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc -- python3 $REPO/tools/inr_bench.py > $OUT/pmc.log 2>&1 || echo "pmc failed" >> $OUT/errors.log
# effective clock = GRBM_GUI_ACTIVE / 8 / kernel time (guide: DVFS give-back); LDS stalls
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc2 -- python3 $REPO/tools/inr_bench.py > $OUT/pmc2.log 2>&1 || echo "pmc2 failed" >> $OUT/errors.log
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + '/trace/*/*_kernel_stats.csv'):
    for r in list(csv.DictReader(open(f)))[:5]: print(r['Name'][:80], r['Calls'], r['AverageNs'], r['Percentage'])
agg = collections.defaultdict(list)
for f in glob.glob(out + '/pmc*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'inr_forward' in r['Kernel_Name']:
            agg[(r['Kernel_Name'][:48], r['Counter_Name'])].append(float(r['Counter_Value']))
for k in sorted(agg): print(k, f"n={len(agg[k])} mean={sum(agg[k])/len(agg[k]):.5g}")
PY
