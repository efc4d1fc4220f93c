#!/usr/bin/env python3
"""Interleaved A/B timing of K1 configurations in ONE process (guide rule 24): N variants x M
rounds, median and min kernel time from HIP events on the launch stream.  Not the headline
bench (that is bench.py) — a development tool whose output goes under gpurun_out/."""
import argparse, itertools, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mrirt
from mrirt import synth

ap = argparse.ArgumentParser()
ap.add_argument("--volume", type=int, default=512)
ap.add_argument("--image", type=int, default=1024)
ap.add_argument("--march-steps", type=int, default=512)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--alpha", type=float, default=16.0)
ap.add_argument("--layouts", default="brick,linear")
ap.add_argument("--maths", default="strict,fast")
ap.add_argument("--shades", default="1,0")
ap.add_argument("--variants", default="0")
ap.add_argument("--cameras", default="80:25", help="phi:theta[,phi:theta...] degrees")
a = ap.parse_args()

n = a.volume
vol = synth.synth_volume(n)
grids = {l: mrirt.upload_grid(vol, (n, n, n), l) for l in a.layouts.split(",")}
torch.cuda.synchronize()
print("grid GiB:", {l: round(g.nbytes / 2**30, 2) for l, g in grids.items()})
cases = []
for cam in a.cameras.split(","):
    phi, th = (float(v) for v in cam.split(":"))
    p = synth.brats_scene(n, a.image, a.march_steps, channels=1, intensity_alpha=a.alpha,
                          camera=synth.bench_camera(phi_deg=phi, theta_deg=th))
    for layout, math_, shade, var in itertools.product(a.layouts.split(","), a.maths.split(","), a.shades.split(","), a.variants.split(",")):
        if layout == "quad" and shade == "1":
            continue
        ext = dict(synth.SHADE_EXT) if shade == "1" else {}
        ext.update(math=math_, layout=layout, kernelVariant=int(var))
        cases.append(dict(cam=cam, layout=layout, math=math_, shade=shade, variant=int(var), p=p, ext=ext, ms=[]))
out = torch.empty((a.image, a.image, 4), device="cuda")
ref_img = {}
for c in cases:   # warm + counts + cross-variant image check
    img, st = mrirt.render_brats(c["p"], [grids[c["layout"]]], out=out, ext=c["ext"], stats=True)
    c.update(st)
    key = (c["cam"], c["shade"], c["math"])
    if key in ref_img:
        c["maxabs_vs_first"] = float((img - ref_img[key]).abs().max())
    else:
        ref_img[key] = img.clone(); c["maxabs_vs_first"] = 0.0
for r in range(a.rounds):
    for c in cases:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); mrirt.render_brats(c["p"], [grids[c["layout"]]], out=out, ext=c["ext"]); e1.record()
        torch.cuda.synchronize()
        c["ms"].append(e0.elapsed_time(e1))
print(f"{'cam':>8} {'layout':>6} {'math':>6} {'sh':>2} {'var':>3} {'med ms':>8} {'min ms':>8} {'Gsamp/s':>8} {'alg GB/s':>9} {'frac':>6} {'dimg':>9}")
for c in cases:
    med, mn = float(np.median(c["ms"])), float(np.min(c["ms"]))
    by = c["live_samples"] * 32 + c["shaded_samples"] * 192 + a.image * a.image * 16
    print(f"{c['cam']:>8} {c['layout']:>6} {c['math']:>6} {c['shade']:>2} {c['variant']:>3} {med:8.3f} {mn:8.3f} "
          f"{c['live_samples'] / med / 1e6:8.2f} {by / med / 1e6:9.1f} {by / med / 1e6 / 8000:6.3f} {c['maxabs_vs_first']:9.2e}")
    c.pop("p"); c.pop("ext")
json.dump(cases, open(os.environ.get("SWEEP_JSON", "/tmp/sweep.json"), "w"))
