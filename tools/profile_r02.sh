#!/bin/bash
# rocprofv3 evidence for one bench.py configuration: kernel trace + stats, then one PMC pass per counter group
# (FETCH_SIZE and WRITE_SIZE need separate passes: TCC has 4 slots — MI355X_MICROARCH.md), then the
# profiles/traffic.json entry (tools/make_traffic.py, keyed by <key>, tagged with the source digest).
# Usage (on the GPU box, from the repo root):
#   bash tools/profile_r02.sh <tag> <traffic key> <kernel substr> [bench args...]
# Output: gpurun_out/prof_<tag>/{trace,pmc_*}/..., gpurun_out/prof_<tag>/traffic_entry.json, and
#         gpurun_out/profiles_<tag>/ (the CSVs to commit under profiles/)
set -o pipefail
TAG=$1; KEY=$2; KSUB=$3; shift 3
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-inr "$@" > $OUT/bench_trace.log 2>&1 || { tail -20 $OUT/bench_trace.log; exit 1; }
echo "trace done"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-60)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-inr "$@" > $OUT/pmc_$N.log 2>&1 || echo "pmc $C failed" >> $OUT/errors.log
  echo "pmc $C done"
done
cd $REPO
python3 tools/make_traffic.py $OUT "$KEY" --kernel "$KSUB" --json $OUT/traffic_entry.json --copy-to $REPO/gpurun_out/profiles_$TAG --source-label "profiles/r02_$TAG/*.csv" | tail -40
grep -h '^{' $OUT/bench_trace.log | tail -1 > $REPO/gpurun_out/profiles_$TAG/bench_line_under_rocprof.json
