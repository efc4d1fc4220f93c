import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import mrirt
from mrirt import synth
n, image = 512, 1024
vol = synth.synth_volume(n)
g = mrirt.upload_grid(vol, (n, n, n), "vga")
p = synth.brats_scene(n, image, 512, channels=1, intensity_alpha=16.0)
ext = dict(synth.SHADE_EXT, layout="vga")
_, s0 = mrirt.render_brats(p, [g], ext=dict(ext, kernelVariant=64), stats=True)
_, s1 = mrirt.render_brats(p, [g], ext=dict(ext, kernelVariant=64 + 128), stats=True)
_, s2 = mrirt.render_brats(p, [g], ext=dict(ext, kernelVariant=64 + 256), stats=True)
print("ring miss fraction", (s2["shaded_samples"] - s0["shaded_samples"]) / s0["live_samples"])
print("live", s0["live_samples"], "shaded", s0["shaded_samples"], "LDS-served", s1["shaded_samples"] - s0["shaded_samples"],
      "fraction", (s1["shaded_samples"] - s0["shaded_samples"]) / s0["live_samples"])
