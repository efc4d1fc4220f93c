import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2] / "tests"))
import numpy as np, torch, mrirt
from mrirt import synth
from oracle import oracle_c as oc
from test_gpu_labcell import _labels
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(8800 + seed)
dims = tuple(int(v) for v in rng.integers(14, 48, 3))
nmod = 1 + seed % 4
vols = [synth.synth_volume(0, 60 + 5 * seed + m, phase=0.4 * m, dims=dims) for m in range(4)]
seg, pred = _labels(rng, dims, 9), _labels(rng, dims, 7)
cam = synth.bench_camera(radius=float(rng.choice([0.3, 1.1, 2.4, 3.2])), phi_deg=float(rng.uniform(8, 172)), theta_deg=float(rng.uniform(0, 360)))
show_seg, show_pred = [(True, True), (True, False), (False, True)][seed % 3]
p = synth.brats_scene(0, 0, int(rng.integers(30, 220)), dims=dims, image_hw=(int(rng.integers(9, 80)), int(rng.integers(9, 80))),
                      channels=nmod, show_seg=show_seg, show_pred=show_pred, intensity_alpha=float(rng.choice([0.4, 6.0])),
                      camera=cam, fov_deg=float(rng.uniform(25, 80)))
p["volWeight"] = tuple(float(v) for v in rng.uniform(0.3, 1.4, 4))
p["gamma"] = float(rng.choice([1.0, 1.0, 1.6]))
print("dims", dims, "nmod", nmod, "seg/pred", show_seg, show_pred, "gamma", p["gamma"], "img", p["imageSize"])
ref = oc.brats_main(p, vols, seg if show_seg else None, pred if show_pred else None, None)
grids = [mrirt.upload_grid(v, dims, "quad") for v in vols]
cells = mrirt.upload_label_cells(seg if show_seg else None, pred if show_pred else None, dims)
gl, gp = mrirt.upload_grid(seg, dims, "brick"), mrirt.upload_grid(pred, dims, "brick")
for name, kw in (("bricks pipelined", dict(labels=gl if show_seg else None, preds=gp if show_pred else None, ext=dict(layout="quad"))),
                 ("cells pipelined", dict(labels=cells, ext=dict(layout="quad"))),
                 ("cells generic", dict(labels=cells, ext=dict(layout="quad", kernelVariant=4))),
                 ("cells skip", dict(labels=cells, ext=dict(layout="quad"), skip=True))):
    got = mrirt.render_brats(p, grids, **kw).cpu().numpy()
    d = np.abs(got - ref).max(axis=-1)
    print(f"{name:18s} differing px {int((d > 0).sum()):6d} of {d.size}  max {d.max():.3e}")
