#!/usr/bin/env python3
"""Exact empty-space skipping on a volume that has empty space: the C3 geometry (512^3, 1024^2, 512 steps,
gradient shading + ERT) on a synthetic skull-stripped head (textured ellipsoid in zeros), with and without
skip=True.  The frames are compared bit for bit; the synthetic bench volume of bench.py has no empty space, so
this is the number that says what skipping buys on scan-like data."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mrirt
from mrirt import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
image, steps = 2 * n, n
rng = np.random.default_rng(5)
ax = np.linspace(-1, 1, n, dtype=np.float32)
z, y, x = np.meshgrid(ax, ax, ax, indexing="ij")
r = np.sqrt((x / 0.62) ** 2 + (y / 0.78) ** 2 + (z / 0.66) ** 2)           # head-shaped ellipsoid: ~17 % of the box
vol = (np.clip(1.05 - r, 0, 1) * (0.75 + 0.25 * np.sin(9 * x) * np.cos(7 * y) * np.sin(6 * z))).astype(np.float32)
vol += 0.04 * rng.random((n, n, n), dtype=np.float32)
vol[r > 1.0] = 0.0
vol = np.ascontiguousarray(vol).reshape(-1)
print(f"occupied voxels: {float((vol > 0).mean()):.3f}")
for layout, shade in (("vga", True), ("vg", True), ("quad", False)):
    g = mrirt.upload_grid(vol, (n, n, n), layout)
    for alpha in (16.0, 2.0):
        p = synth.brats_scene(n, image, steps, channels=1, intensity_alpha=alpha)
        ext = dict(synth.SHADE_EXT) if shade else {}
        ext.update(layout=layout)
        out = torch.empty((image, image, 4), device="cuda")
        res = {}
        for mode in ("plain", "level1", "skip"):            # level1: one step at a time (kernelVariant bit 8); skip: + packet leaps
            e = dict(ext, kernelVariant=256) if mode == "level1" else ext
            skip = mode != "plain"
            for _ in range(3):
                mrirt.render_brats(p, [g], out=out, ext=e, skip=skip)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                mrirt.render_brats(p, [g], out=out, ext=e, skip=skip)
            e1.record(); torch.cuda.synchronize()
            _, st = mrirt.render_brats(p, [g], out=out, ext=e, skip=skip, stats=True)
            res[mode] = (e0.elapsed_time(e1) / 10, out.clone(), st)
        same = all(torch.equal(res["plain"][1], res[m][1]) and res["plain"][2] == res[m][2] for m in ("level1", "skip"))
        st = res["skip"][2]
        print(f"{n}^3 {image}^2 x {steps}  {layout:4s} shade={int(shade)} alpha={alpha:4.1f}: plain {res['plain'][0]:.3f} ms, "
              f"8^3 cells, step by step {res['level1'][0]:.3f} ms, + distance-map leaps {res['skip'][0]:.3f} ms ({res['plain'][0] / res['skip'][0]:.2f}x), "
              f"live {st['live_samples'] / 1e6:.1f} M / shaded {st['shaded_samples'] / 1e6:.1f} M, identical={same}", flush=True)
