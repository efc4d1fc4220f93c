#!/bin/bash
# rocprofv3 evidence for one configuration: kernel trace + stats, then one PMC pass per counter group (FETCH_SIZE and
# WRITE_SIZE need separate passes: the TCC block has 4 slots — MI355X_MICROARCH.md), then the profiles/traffic.json entry
# (tools/make_traffic.py: keyed by <key>, tagged with the digest of the kernel sources).
# Usage (on the GPU box, from the repo root):
#   bash tools/profile_r04.sh <tag> <traffic key> <kernel substr> <python script> [script args...]
# e.g. bash tools/profile_r04.sh c3_vga "C3:512:1024:512:vga:strict:shade" "pipe_kernel<true, 4" bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-inr --no-k1
# Output: gpurun_out/prof_<tag>/..., gpurun_out/profiles_<tag>/ (the CSVs to commit under profiles/r04_<tag>/)
set -o pipefail
TAG=$1; KEY=$2; KSUB=$3; SCRIPT=$4; shift 4
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT $REPO/gpurun_out/profiles_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/$SCRIPT "$@" > $OUT/run_trace.log 2>&1 || { tail -20 $OUT/run_trace.log; exit 1; }
echo "trace done"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TA_TA_BUSY_sum TD_TD_BUSY_sum" "GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-60)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- python3 $REPO/$SCRIPT "$@" > $OUT/pmc_$N.log 2>&1 || echo "pmc $C failed" >> $OUT/errors.log
  echo "pmc $C done"
done
cd $REPO
python3 tools/make_traffic.py $OUT "$KEY" --kernel "$KSUB" --json $OUT/traffic_entry.json --copy-to $REPO/gpurun_out/profiles_$TAG --source-label "profiles/r04_$TAG/*.csv" | tail -45
grep -h '^{' $OUT/run_trace.log | tail -1 > $REPO/gpurun_out/profiles_$TAG/line_under_rocprof.json
grep -h '^C2 variant' $OUT/run_trace.log | tail -1 >> $REPO/gpurun_out/profiles_$TAG/line_under_rocprof.json
exit 0
