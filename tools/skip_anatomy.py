#!/usr/bin/env python3
"""Where a skipping frame's time goes: the C3 geometry on (a) an all-zero volume (every step empty), (b) a dense volume
(no step empty), (c) the synthetic head — plain launch, 8^3 cells only (kernelVariant 256), 8^3 + 32^3 leaps."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mrirt
from mrirt import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
image, steps = 2 * n, n
ax = np.linspace(-1, 1, n, dtype=np.float32)
z, y, x = np.meshgrid(ax, ax, ax, indexing="ij")
r = np.sqrt((x / 0.62) ** 2 + (y / 0.78) ** 2 + (z / 0.66) ** 2)
head = (np.clip(1.05 - r, 0, 1) * (0.75 + 0.25 * np.sin(9 * x) * np.cos(7 * y) * np.sin(6 * z))).astype(np.float32)
head[r > 1.0] = 0.0
cases = {"zeros": np.zeros(n ** 3, np.float32), "dense": synth.synth_volume(n), "head": np.ascontiguousarray(head).reshape(-1)}
del x, y, z, r
for name, vol in cases.items():
    for layout, shade in (("vga", True), ("quad", False)):
        g = mrirt.upload_grid(vol, (n, n, n), layout)
        p = synth.brats_scene(n, image, steps, channels=1, intensity_alpha=16.0)
        ext = dict(synth.SHADE_EXT) if shade else {}
        ext.update(layout=layout)
        out = torch.empty((image, image, 4), device="cuda")
        line = f"{name:6s} {layout:4s}:"
        for mode in ("plain", "level1", "skip"):
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
            line += f"  {mode} {e0.elapsed_time(e1) / 10:.3f} ms"
        print(line + f"   live {st['live_samples'] / 1e6:.1f} M shaded {st['shaded_samples'] / 1e6:.1f} M", flush=True)
        del g
