#!/bin/bash
# builds and runs tools/micro/tcp_tag_rate.hip, then once more under rocprofv3 for the TCP look-ups per load
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/tcp_tag_rate
rm -rf $OUT; mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $REPO/tools/micro/tcp_tag_rate.hip -o /tmp/tcp_tag_rate || exit 1
/tmp/tcp_tag_rate 4096 | tee $OUT/timing.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc -- /tmp/tcp_tag_rate 4096 > $OUT/pmc.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_BUSY_CU_CYCLES TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $OUT/pmc2 -- /tmp/tcp_tag_rate 4096 > $OUT/pmc2.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(dict)
for f in glob.glob(out + '/pmc*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        # two dispatches per pattern (warm-up, timed): keep the larger (timed) one
        k = r['Kernel_Name']
        v = float(r['Counter_Value'])
        agg[k][r['Counter_Name']] = max(agg[k].get(r['Counter_Name'], 0.0), v)
with open(out + '/counters.txt', 'w') as fh:
    for k in sorted(agg):
        c = agg[k]
        loads = c.get('SQ_INSTS_VMEM_RD', 0.0)
        acc = c.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0.0)
        gui = c.get('GRBM_GUI_ACTIVE', 0.0) / 8.0        # the counter is summed over the 8 XCDs
        line = (f"{k[:48]:48s} wave-level loads {loads:.4g}  TCP accesses {acc:.4g}  = {acc / loads if loads else 0:.2f} per load;  "
                f"GRBM_GUI_ACTIVE {gui:.4g} -> {acc / gui / 256 if gui else 0:.3f} accesses/clk/CU, {loads / gui / 256 if gui else 0:.4f} loads/clk/CU;  "
                f"L2 read requests {c.get('TCP_TCC_READ_REQ_sum', 0.0):.4g}")
        print(line); fh.write(line + "\n")
PY
