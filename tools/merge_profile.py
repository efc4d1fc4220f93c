#!/usr/bin/env python3
"""After `gpurun -- bash tools/profile_r02.sh <tag> <key> <kernel>` (+ optionally a `rocprofv3 --kernel-trace --stats`
of the default bench command into gpurun_out/prof_bench_default): merge the traffic entry into profiles/traffic.json
and copy the CSV summaries into profiles/r02_<tag>/ and profiles/r02_bench_default/.
    python3 tools/merge_profile.py <tag>"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
e = json.load(open(f"{ROOT}/gpurun_out/prof_{tag}/traffic_entry.json"))
path = f"{ROOT}/profiles/traffic.json"
db = json.load(open(path))
db.update(e)
with open(path, "w") as fh:
    json.dump(db, fh, indent=2); fh.write("\n")
dst = f"{ROOT}/profiles/r02_{tag}"
os.makedirs(dst, exist_ok=True)
for f in glob.glob(f"{ROOT}/gpurun_out/profiles_{tag}/*"):
    shutil.copy(f, dst)
src = f"{ROOT}/gpurun_out/prof_bench_default"
if os.path.isdir(src):
    d2 = f"{ROOT}/profiles/r02_bench_default"
    os.makedirs(d2, exist_ok=True)
    for f in glob.glob(f"{src}/*/*kernel_stats.csv") + glob.glob(f"{src}/*/*domain_stats.csv"):
        shutil.copy(f, f"{d2}/" + os.path.basename(f).split("_", 1)[1])
    log = f"{ROOT}/gpurun_out/prof_bench_default.log"
    if os.path.exists(log):
        lines = [l for l in open(log) if l.startswith("{")]
        if lines:
            open(f"{d2}/bench_line_under_rocprof.json", "w").write(lines[-1])
print({k: (v.get("source_digest"), v.get("hbm_bytes_per_launch"), v.get("kernel_ms_under_profiler"), v.get("on_chip")) for k, v in e.items() if isinstance(v, dict)})
