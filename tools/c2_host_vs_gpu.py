import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, mrirt
from mrirt import synth
n, image, steps = 256, 512, 256
vols = [synth.synth_volume(n, 1234 + m, phase=0.3 * m) for m in range(4)]
gv = [mrirt.upload_grid(v, (n, n, n), "quad") for v in vols]
gl = mrirt.upload_grid(synth.synth_labels(n), (n, n, n), "brick")
p = synth.brats_scene(n, image, steps, channels=4, show_seg=True, intensity_alpha=0.4)
out = torch.empty((image, image, 4), device="cuda")
ext = dict(layout="quad")
for _ in range(5): mrirt.render_brats(p, gv, labels=gl, out=out, ext=ext)
torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
for _ in range(N): mrirt.render_brats(p, gv, labels=gl, out=out, ext=ext)
th = (time.perf_counter() - t0) / N
torch.cuda.synchronize()
ta = (time.perf_counter() - t0) / N
print(f"host per call {th*1e3:.3f} ms, incl. GPU drain {ta*1e3:.3f} ms")
for variant in (0, 2, 512):
    e = dict(ext, kernelVariant=variant)
    for _ in range(3): mrirt.render_brats(p, gv, labels=gl, out=out, ext=e)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N): mrirt.render_brats(p, gv, labels=gl, out=out, ext=e)
    e1.record(); torch.cuda.synchronize()
    print("variant", variant, e0.elapsed_time(e1) / N)
