"""Config 2 (256^3 x 4 + seg, 512^2, 256 steps, QUAD, STRICT): brick label grid vs label cells."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np, torch, mrirt
from mrirt import synth
n = 256
vols = [synth.synth_volume(n, 1234 + m, phase=0.3 * m) for m in range(4)]
lab = synth.synth_labels(n)
p = synth.brats_scene(n, 512, 256, channels=4, show_seg=True, intensity_alpha=0.4)
grids = [mrirt.upload_grid(v, (n, n, n), "quad") for v in vols]
gl = mrirt.upload_grid(lab, (n, n, n), "brick")
gc = mrirt.upload_label_cells(lab, None, (n, n, n))
out = torch.empty((512, 512, 4), dtype=torch.float32, device="cuda")
def t(labels, reps=40):
    for _ in range(5): mrirt.render_brats(p, grids, labels=labels, out=out)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): mrirt.render_brats(p, grids, labels=labels, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for i in range(3):
    print(f"brick labels {t(gl):.4f} ms   label cells {t(gc):.4f} ms")
