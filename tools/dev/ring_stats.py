"""Ring kernel diagnostics on the bench geometry: rounds per wave, idle lane-rounds, uncovered reads."""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np, torch
import mrirt
from mrirt import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
image, steps = (1024, 512) if n == 512 else (2 * n, n)
vol = synth.synth_volume(n, 1234)
g = mrirt.upload_grid(vol, (n, n, n), "vga")
p = synth.brats_scene(n, image, steps, channels=1, intensity_alpha=16.0)
ext = dict(synth.SHADE_EXT, layout="vga", math="strict")
def run(v):
    return mrirt.render_brats(p, [g], ext=dict(ext, kernelVariant=v), stats=True)
f0, s0 = run(0)
f1, s1 = run(2048)
print("equal frames", torch.equal(f0, f1), "stats equal", s0 == s1, s0)
live = s1["live_samples"]
_, s = run(2048 | 128); print("uncovered + fallback samples", s["shaded_samples"] - s1["shaded_samples"], f"({(s['shaded_samples'] - s1['shaded_samples']) / live:.4f} of live)")
_, s = run(2048 | 512); idle = s["shaded_samples"] - s1["shaded_samples"]; print("idle lane-rounds", idle, f"({idle / live:.4f} of live)")
_, s = run(2048 | 512 | 128); rounds = s["shaded_samples"] - s1["shaded_samples"]; print("wave-rounds", rounds, "vs live/64 =", live / 64, "ratio", rounds / (live / 64))
def t(v, reps=20):
    run(v); torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(reps): mrirt.render_brats(p, [g], ext=dict(ext, kernelVariant=v))
    torch.cuda.synchronize(); return (time.perf_counter() - a) / reps * 1e3
for name, v in (("default", 0), ("8x8 gather", 2), ("ring", 2048), ("ring R=3", 2048 | 4096), ("ring no loader", 2048 | 256), ("ring R=3 no loader", 2048 | 4096 | 256), ("ring all fallback", 2048 | 1024)):
    print(f"{name:18s} {t(v):.3f} ms")
