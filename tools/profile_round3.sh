#!/bin/bash
# The round-3 evidence set, one GPU session: PMC + trace for the two march kernels bench.py reports (-> traffic.json entries),
# rocprofv3 stats of the default bench command, then the secondary benchmarks as text.  Run from the repo root on the GPU box:
#   bash tools/profile_round3.sh            (outputs under gpurun_out/r03/ and gpurun_out/profiles_*/)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r03
rm -rf $O; mkdir -p $O
cd $REPO
bash tools/profile_r03.sh c3_vga "C3:512:1024:512:vga:strict:shade" "pipe_kernel<true, 4" bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-inr --no-k1 > $O/profile_c3.log 2>&1; echo "c3 pmc done"
bash tools/profile_r03.sh c2_quad "C2:256:512:256:quad:strict:4ch+seg" "pipe_kernel<true, 3, false, 4" tools/c2_run.py 10 > $O/profile_c2.log 2>&1; echo "c2 pmc done"
python3 - <<'PY'
import json, glob, os
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
db_path = os.path.join(root, "profiles", "traffic.json")
db = json.load(open(db_path))
for tag in ("c3_vga", "c2_quad"):
    f = os.path.join(root, "gpurun_out", f"prof_{tag}", "traffic_entry.json")
    if os.path.exists(f):
        for k, v in json.load(open(f)).items():
            if not k.startswith("_"):
                db[k] = v
json.dump(db, open(os.path.join(root, "gpurun_out", "r03", "traffic.json"), "w"), indent=2)
PY
cp $O/traffic.json $REPO/profiles/traffic.json          # (the copy in this box's tree: bench.py below reads it)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_default_trace -- python3 $REPO/bench.py > $O/bench_default_under_rocprof.json 2> $O/bench_default_under_rocprof.err ); echo "bench trace done"
python3 - <<'PY'
import csv, glob, os
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
o = os.path.join(root, "gpurun_out", "r03")
for f in glob.glob(o + "/bench_default_trace/*/*_kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    with open(o + "/bench_default_kernel_stats.csv", "w") as fh:
        w = csv.DictWriter(fh, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
PY
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench done"
python3 bench.py --force-exchange --image 2048 --steps 10 --warmup 2 --no-inr --no-k1 --no-cpu-baseline > $O/bench_force_exchange.json 2>/dev/null; echo "bench fe done"
python3 tools/configs_bench.py > $O/configs_bench.txt 2>/dev/null; echo "configs done"
python3 tools/c5_bench.py > $O/c5_bench.txt 2>/dev/null
python3 tools/tile_share_bench.py 0 2 > $O/tile_share_bench.txt 2>/dev/null
python3 tools/viewer_frame_bench.py > $O/viewer_frame_bench.txt 2>/dev/null
python3 tools/multimod_bench.py > $O/multimod_bench.txt 2>/dev/null
python3 tools/skip_bench.py > $O/skip_bench.txt 2>/dev/null
python3 tools/inr_refine_check.py 400000 > $O/inr_refine_check.txt 2>/dev/null
MRIRT_INR_NO_REFINE=1 python3 tools/inr_bench.py > $O/inr_bench_bf16_pass_only.txt 2>/dev/null
python3 tools/inr_bench.py > $O/inr_bench_with_refinement.txt 2>/dev/null
python3 tools/k2_bench.py > $O/k2_bench.txt 2>/dev/null
bash tools/profile_c5.sh r03 > /dev/null 2>&1; cp $REPO/gpurun_out/prof_c5_r03/kernel_stats.txt $O/c5_frame_kernel_stats.txt 2>/dev/null
echo "all done"; ls $O
