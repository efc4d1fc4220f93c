#!/bin/bash
# PMC passes for one bench configuration.  Usage: bash tools/pmc.sh <tag> "<counter set 1>" "<set 2>" ... -- <bench args>
TAG=$1; shift
SETS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do SETS+=("$1"); shift; done
shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/set$i -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-inr "$@" > $OUT/set$i.log 2>&1 || echo "set $i ($C) failed" >> $OUT/errors.log
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + '/set*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'march' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/summary.txt', 'w') as fh:
    for k in sorted(agg):
        line = f"{k:36s} n={len(agg[k])} mean={sum(agg[k])/len(agg[k]):.5g}"
        print(line); fh.write(line + "\n")
PY
