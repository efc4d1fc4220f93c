#!/usr/bin/env python3
"""The viewer's shaded four-modality frame (256^3 x 4 VG grids, 512^2 or 1024^2, 256 steps, seg overlay): the rolling
per-modality pipeline (default) against the generic kernel (kernelVariant 4); frames must be the same bits."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mrirt
from mrirt import synth
n = 256
vols = [synth.synth_volume(n, 1234 + m, phase=0.3 * m) for m in range(4)]
lab = synth.synth_labels(n)
gl = mrirt.upload_grid(lab, (n, n, n), "brick")
for layout in ("vg", "vga"):
    gv = [mrirt.upload_grid(v, (n, n, n), layout) for v in vols]
    for image in (512, 1024):
        for ch, seg in ((4, True), (4, False), (2, False)):
            p = synth.brats_scene(n, image, 256, channels=ch, show_seg=seg, intensity_alpha=4.0)
            ext = dict(synth.SHADE_EXT)
            out = {}
            for variant in (0, 4):
                e = dict(ext, kernelVariant=variant)
                img, st = mrirt.render_brats(p, gv, gl if seg else None, ext=e, stats=True)
                ts = []
                for _ in range(7):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); mrirt.render_brats(p, gv, gl if seg else None, ext=e); e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                out[variant] = (img, float(np.median(ts)), st["live_samples"])
            same = torch.equal(out[0][0], out[4][0])
            print(f"{layout:4s} {image}^2 {ch} modalities seg={int(seg)}: rolling {out[0][1]:.3f} ms ({out[0][2] * ch / out[0][1] / 1e6:.1f} G modality-samples/s), "
                  f"generic {out[4][1]:.3f} ms, same bits: {same}", flush=True)
    del gv
