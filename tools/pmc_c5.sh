#!/bin/bash
# PMC passes over config-5 frames (tools/c5_bench.py): per-kernel means for the emission / composite / refinement kernels.
# usage: bash tools/pmc_c5.sh <tag> [c5_bench args]
TAG=${1:-r04}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_c5_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" \
         "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TA_TA_BUSY_sum TD_TD_BUSY_sum" "GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-60)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- python3 $REPO/tools/c5_bench.py --nets siren --chunks 96 --frames 2 "$@" > $OUT/pmc_$N.log 2>&1 || echo "pmc $C failed" >> $OUT/errors.log
  echo "pmc $C done"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/pmc_*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        for key in ('c5_emit', 'c5_composite', 'inr_refine', 'c5_plan'):
            if key in k:
                agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/summary.txt', 'w') as fh:
    for key in agg:
        for c in sorted(agg[key]):
            v = [x for x in agg[key][c] if x > 0] or [0.0]
            line = f"{key:14s} {c:36s} n={len(v)} mean={sum(v)/len(v):.6g} max={max(v):.6g}"
            print(line); fh.write(line + "\n")
PY
