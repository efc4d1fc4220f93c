#!/bin/bash
# The round-4 evidence set, one GPU session: PMC + trace for the two march kernels bench.py reports (-> traffic.json entries),
# the 256^3 experiment of VERDICT r3 #5 (same kernel, Infinity-Cache-resident volume), rocprofv3 stats of the default bench
# command, then the secondary benchmarks as text.  Run from the repo root on the GPU box:
#   bash tools/profile_round4.sh            (outputs under gpurun_out/r04/ and gpurun_out/profiles_*/)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r04
rm -rf $O; mkdir -p $O
cd $REPO
bash tools/profile_r04.sh c3_vga "C3:512:1024:512:vga:strict:shade" "pipe_kernel<true, 4" bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-inr --no-k1 --no-scaling-model --no-pipelined > $O/profile_c3.log 2>&1; echo "c3 pmc done"
bash tools/profile_r04.sh c2_mod4 "C2:256:512:256:mod4:strict:4ch+seg" "pipe_kernel<true, 6, false, 4" tools/c2_run.py 10 0 mod4 > $O/profile_c2.log 2>&1; echo "c2 (mod4) pmc done"
bash tools/profile_r04.sh c2_quad "C2:256:512:256:quad:strict:4ch+seg" "pipe_kernel<true, 3, false, 4" tools/c2_run.py 10 0 quad > $O/profile_c2q.log 2>&1; echo "c2 (quad) pmc done"
# the C3 kernel on a 256^3 volume (1.5 GiB of VGA voxels: Infinity-Cache resident), at the same ray spacing in voxels
# (512^2 px, 256 steps) and at the full image (1024^2 px, 512 steps)
bash tools/profile_r04.sh c3_256_512px "C3:256:512:256:vga:strict:shade" "pipe_kernel<true, 4" bench.py --volume 256 --image 512 --march-steps 256 --steps 10 --warmup 2 --no-cpu-baseline --no-inr --no-k1 --no-scaling-model --no-pipelined > $O/profile_c3_256a.log 2>&1; echo "c3 256 (512px) pmc done"
bash tools/profile_r04.sh c3_256_1024px "C3:256:1024:512:vga:strict:shade" "pipe_kernel<true, 4" bench.py --volume 256 --steps 10 --warmup 2 --no-cpu-baseline --no-inr --no-k1 --no-scaling-model --no-pipelined > $O/profile_c3_256b.log 2>&1; echo "c3 256 (1024px) pmc done"
python3 - <<'PY'
import json, glob, os
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
db_path = os.path.join(root, "profiles", "traffic.json")
db = json.load(open(db_path))
for tag in ("c3_vga", "c2_mod4", "c2_quad", "c3_256_512px", "c3_256_1024px"):
    f = os.path.join(root, "gpurun_out", f"prof_{tag}", "traffic_entry.json")
    if os.path.exists(f):
        for k, v in json.load(open(f)).items():
            if not k.startswith("_"):
                db[k] = v
json.dump(db, open(os.path.join(root, "gpurun_out", "r04", "traffic.json"), "w"), indent=2)
PY
cp $O/traffic.json $REPO/profiles/traffic.json          # (the copy in this box's tree: bench.py below reads it)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_default_trace -- python3 $REPO/bench.py > $O/bench_default_under_rocprof.json 2> $O/bench_default_under_rocprof.err ); echo "bench trace done"
python3 - <<'PY'
import csv, glob, os
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
o = os.path.join(root, "gpurun_out", "r04")
for f in glob.glob(o + "/bench_default_trace/*/*_kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    with open(o + "/bench_default_kernel_stats.csv", "w") as fh:
        w = csv.DictWriter(fh, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
PY
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench done"
python3 bench.py --force-exchange --image 2048 --steps 10 --warmup 2 --no-inr --no-k1 --no-cpu-baseline > $O/bench_force_exchange.json 2>/dev/null; echo "bench fe done"
python3 tools/configs_bench.py > $O/configs_bench.txt 2>/dev/null; echo "configs done"
python3 tools/c5_bench.py > $O/c5_bench.txt 2>/dev/null
python3 tools/viewer_frame_bench.py > $O/viewer_frame_bench.txt 2>/dev/null
python3 tools/refine_bench.py > $O/refine_bench.txt 2>/dev/null
bash tools/profile_c5.sh r04 --chunks 96 > $O/c5_frame_kernel_stats.txt 2>&1
echo "all done"
