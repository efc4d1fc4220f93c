"""What the two label overlays cost in the viewer's frame (QUAD 4 modalities, 1280x720, 240x240x155)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, mrirt
from mrirt import synth
dims = (240, 240, 155); W, H = 1280, 720
vols = [synth.synth_volume(0, 1234 + m, phase=0.3 * m, dims=dims) for m in range(4)]
lab = synth.synth_labels(0, dims=dims); pred = np.roll(lab, 3).copy()
g = [mrirt.upload_grid(v, dims, "quad") for v in vols]
gl, gp = mrirt.upload_grid(lab, dims, "brick"), mrirt.upload_grid(pred, dims, "brick")
out = torch.empty((H, W, 4), dtype=torch.float16, device="cuda")
for seg, prd in ((1, 1), (1, 0), (0, 0)):
    p = synth.brats_scene(0, 0, 64, dims=dims, image_hw=(H, W), channels=4, show_seg=bool(seg), show_pred=bool(prd), intensity_alpha=0.4)
    p["stepSize"] = np.float32(0.05)
    ext = dict(layout="quad", labelLayout="brick", outFormat="rgba16f")
    def run():
        mrirt.render_brats(p, g, labels=gl if seg else None, preds=gp if prd else None, ext=ext, out=out)
    for _ in range(5): run()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
    for a, b in ev:
        a.record(); run(); b.record()
    torch.cuda.synchronize()
    print(f"seg={seg} pred={prd}: {np.mean([a.elapsed_time(b) for a, b in ev]):.4f} ms per frame (events around each launch)")
