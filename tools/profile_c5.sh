#!/bin/bash
# rocprofv3 kernel trace of config-5 frames (per-kernel time of plan / emit / MLP / refinement / composite).
# usage: bash tools/profile_c5.sh <tag> [c5_bench args]
TAG=${1:-r03}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_c5_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/c5_bench.py --nets siren --chunks 32 --frames 3 "$@" > $OUT/bench.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in glob.glob(out + '/trace/*/*_kernel_stats.csv'):
    with open(out + '/kernel_stats.txt', 'w') as fh:
        for r in list(csv.DictReader(open(f)))[:14]:
            line = f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:9.3f} ms  {r['Percentage']} %"
            print(line); fh.write(line + "\n")
PY
tail -2 $OUT/bench.log
