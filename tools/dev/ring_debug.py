"""Debug: reproduce tests/test_gpu_ring.py seed N and compare kernels pixel by pixel."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
import mrirt
from mrirt import synth
from oracle import oracle_c as oc
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 15
rng = np.random.default_rng(4100 + seed)
dims = tuple(int(v) for v in rng.integers(17, 60, 3))
vol = synth.synth_volume(0, 20 + seed, phase=float(rng.uniform(0, 3)), dims=dims)
ups = (None, (1.0, 0.0, 0.0), (0.0, 0.0, 1.0))
cam = synth.bench_camera(radius=float(rng.choice([0.2, 0.9, 2.0, 3.0, 5.0])), phi_deg=float(rng.uniform(3, 177)),
                         theta_deg=float(rng.uniform(0, 360)), world_up=None if seed % 3 == 0 else np.array(ups[seed % 3], np.float32))
shade = bool(seed & 1)
p = synth.brats_scene(0, 0, int(rng.choice([24, 60, 150, 400])), dims=dims, image_hw=(int(rng.integers(9, 120)), int(rng.integers(9, 120))),
                      channels=1, intensity_alpha=float(rng.choice([0.4, 16.0, 60.0])), camera=cam, fov_deg=float(rng.uniform(10, 70)))
p["voxelSize"] = (p["voxelSize"] * rng.uniform(0.6, 1.7, 3)).astype(np.float32)
p["gamma"] = float(rng.choice([1.0, 1.0, 1.8]))
if seed % 4 == 3:
    p["volWeight"] = (np.float32(0.7), np.float32(1), np.float32(1), np.float32(1))
ext = dict(synth.SHADE_EXT) if shade else {}
if seed % 5 == 4:
    ext.update(cameraMode=1, orthoHalfHeight=float(rng.uniform(0.3, 1.2)))
OK = ("shadeMode", "ka", "kd", "ks", "specPow2", "gradEps", "cameraMode", "orthoHalfHeight")
print("dims", dims, "gamma", p["gamma"], "alpha", p["intensityAlpha"], "step", p["stepSize"], "img", p["imageSize"], "shade", shade, "w", p["volWeight"])
ref, aux = oc.brats_main(p, [vol], None, None, {k: v for k, v in ext.items() if k in OK}, return_aux=True)
g = mrirt.upload_grid(vol, dims, "vga")
for name, var in (("pipe", 2), ("generic", 4), ("ring", 2048), ("ring+fallback", 2048 | 1024), ("ring+count", 2048 | 128)):
    got, st = mrirt.render_brats(p, [g], ext=dict(ext, math="strict", layout="vga", kernelVariant=var), stats=True)
    d = np.abs(got.cpu().numpy() - ref)[..., 0]
    ys, xs = np.nonzero(d)
    print(f"{name:14s} differing px {len(ys):5d} max {d.max():.3e} live {st['live_samples']} (oracle {aux['live_samples']}) shaded {st['shaded_samples']}", list(zip(ys[:6].tolist(), xs[:6].tolist())))
