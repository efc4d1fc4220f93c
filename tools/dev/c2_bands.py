"""Config 2 with label cells: XCD band height sweep (kernelVariant bits 4-5: 0 = default 8 px, 16 = 16 px, 32 = 32 px, 48 = 64 px; bit 3 = contiguous)."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np, torch, mrirt
from mrirt import synth
n = 256
vols = [synth.synth_volume(n, 1234 + m, phase=0.3 * m) for m in range(4)]
lab = synth.synth_labels(n)
p = synth.brats_scene(n, 512, 256, channels=4, show_seg=True, intensity_alpha=0.4)
grids = [mrirt.upload_grid(v, (n, n, n), "quad") for v in vols]
gc = mrirt.upload_label_cells(lab, None, (n, n, n))
out = torch.empty((512, 512, 4), dtype=torch.float32, device="cuda")
def t(v, reps=40):
    ext = dict(layout="quad", kernelVariant=v)
    for _ in range(5): mrirt.render_brats(p, grids, labels=gc, out=out, ext=ext)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): mrirt.render_brats(p, grids, labels=gc, out=out, ext=ext)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for rep in range(2):
    print("  ".join(f"v{v}: {t(v):.4f}" for v in (0, 16, 32, 48, 8, 512, 512 | 16, 2, 2 | 16)))
