#!/usr/bin/env python3
"""From how many enabled modalities on does the MOD4 grid beat one QUAD grid per modality?  Config-2 geometry (256^3, 512^2, 256 steps,
seg overlay as label cells), 1..4 modalities enabled.    python3 tools/mod4_threshold.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mrirt
from mrirt import synth

n, image, steps = 256, 512, 256
vols = [synth.synth_volume(n, 1234 + m, phase=0.3 * m) for m in range(4)]
gq = [mrirt.upload_grid(v, (n, n, n), "quad") for v in vols]
g4 = mrirt.upload_mod4(vols, (n, n, n))
gl = mrirt.upload_label_cells(synth.synth_labels(n), None, (n, n, n))
out = torch.empty((image, image, 4), device="cuda")


def timed(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for en in ((1, 0, 0, 0), (1, 1, 0, 0), (1, 0, 1, 0), (1, 1, 1, 0), (1, 1, 1, 1)):
    p = synth.brats_scene(n, image, steps, channels=4, show_seg=True, intensity_alpha=0.4)
    p["volEnabled"] = en
    tq = timed(lambda: mrirt.render_brats(p, gq, labels=gl, out=out, ext=dict(layout="quad")))
    t4 = timed(lambda: mrirt.render_brats(p, [g4] * 4, labels=gl, out=out, ext=dict(layout="mod4")))
    a = mrirt.render_brats(p, gq, labels=gl, ext=dict(layout="quad"))
    b = mrirt.render_brats(p, [g4] * 4, labels=gl, ext=dict(layout="mod4"))
    print(f"enabled {en}: one QUAD grid per modality {tq:.3f} ms, MOD4 {t4:.3f} ms, same bits {bool(torch.equal(a, b))}", flush=True)
