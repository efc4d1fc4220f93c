#!/bin/bash
# rocprofv3 evidence for bench.py's default run: kernel trace + stats, then PMC passes
# (FETCH_SIZE and WRITE_SIZE need separate passes: TCC has 4 slots — MI355X_MICROARCH.md).
# Usage (on the GPU box, from the repo root): bash tools/profile_r01.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-inr $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.log 2>&1 || exit 1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-inr "$@" > $OUT/pmc_$N.log 2>&1 || echo "pmc $C failed" >> $OUT/errors.log
done
find $OUT -name '*.csv' | head -50
