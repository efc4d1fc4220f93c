#!/usr/bin/env python3
"""BASELINE config 2 (256^3 x 4 modalities + seg overlay, 512^2, 256 steps, strict) rendered N times: a target
for rocprofv3 (kernel trace / PMC passes).    python3 tools/c2_run.py [frames] [kernelVariant] [mod4|quad]
mod4 (default): the four modalities as one float4 grid (MRIRT_LAYOUT_MOD4); quad: four QUAD grids."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mrirt
from mrirt import synth
n, image, steps = 256, 512, 256
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
layout = sys.argv[3] if len(sys.argv) > 3 else "mod4"
vols = [synth.synth_volume(n, 1234 + m, phase=0.3 * m) for m in range(4)]
gv = [mrirt.upload_mod4(vols, (n, n, n))] * 4 if layout == "mod4" else [mrirt.upload_grid(v, (n, n, n), "quad") for v in vols]
gl = mrirt.upload_label_cells(synth.synth_labels(n), None, (n, n, n))      # as bench.py's k1_reference_path and the shim bind it
p = synth.brats_scene(n, image, steps, channels=4, show_seg=True, intensity_alpha=0.4)
out = torch.empty((image, image, 4), device="cuda")
for _ in range(3):
    mrirt.render_brats(p, gv, labels=gl, out=out, ext=dict(layout=layout, kernelVariant=variant))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(frames):
    mrirt.render_brats(p, gv, labels=gl, out=out, ext=dict(layout=layout, kernelVariant=variant))
e1.record(); torch.cuda.synchronize()
_, st = mrirt.render_brats(p, gv, labels=gl, out=out, ext=dict(layout=layout, kernelVariant=variant), stats=True)
print(f"C2 variant {variant} {layout}: {e0.elapsed_time(e1) / frames:.3f} ms/frame, live {st['live_samples']}")
