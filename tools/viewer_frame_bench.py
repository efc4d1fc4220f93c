#!/usr/bin/env python3
"""The reference viewer's own frame (inr/viewer/brats_viewer.py defaults): a BraTS-sized case, 240 x 240 x 155
voxels, four modalities enabled + GT and prediction overlays, 1280 x 720 window, stepSize 0.05 (the UI default),
gamma 1, through the slangpy-shaped shim exactly as the viewer dispatches it (rgba16_float target)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mrirt
import mrirt.shim as spy
from mrirt import synth

dims = (240, 240, 155)
W, H = 1280, 720
vols = [synth.synth_volume(0, 1234 + m, phase=0.3 * m, dims=dims) for m in range(4)]
lab = synth.synth_labels(0, dims=dims)
pred = np.roll(lab, 3).copy()
p = synth.brats_scene(0, 0, 64, dims=dims, image_hw=(H, W), channels=4, show_seg=True, show_pred=True, intensity_alpha=0.4)
p["stepSize"] = np.float32(0.05)
for layout, skip in (("auto", True), ("auto", False), ("linear", True)):
    dev = spy.Device(layout=layout, skip_empty=skip)
    k = dev.create_compute_kernel(dev.load_program("brats_rt.slang", ["brats_main"]))
    bufs = []
    for a in vols + [lab, pred]:
        b = dev.create_buffer(element_count=a.size, struct_size=4)
        b.copy_from_numpy(a)
        bufs.append(b)
    tex = dev.create_texture(format=spy.Format.rgba16_float, width=W, height=H)
    vars_ = {"gOutput": tex, "gParams": p, "gLabels": bufs[4], "gPreds": bufs[5], **{f"gIntensity{m}": bufs[m] for m in range(4)}}
    for _ in range(3):
        k.dispatch(thread_count=[W, H, 1], vars=vars_)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        k.dispatch(thread_count=[W, H, 1], vars=vars_)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(f"viewer frame {W}x{H}, {dims}, 4 modalities + seg + pred, stepSize 0.05, Device(layout={layout!r}, skip_empty={skip}): {ms:.3f} ms per dispatch incl. Python ({1000 / ms:.0f} frames/s)")
