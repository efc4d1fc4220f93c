#!/usr/bin/env python3
"""Where a viewer frame's time goes: the host side of ComputeKernel.dispatch (cProfile) against the GPU time of the same
frame (kernel-only HIP events around the C call).    python3 tools/viewer_dispatch_profile.py"""
import cProfile, io, os, pstats, sys, time
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
dev = spy.Device(layout="auto")
k = dev.create_compute_kernel(dev.load_program("brats_rt.slang", ["brats_main"]))
bufs = []
for a in vols + [lab, pred]:
    b = dev.create_buffer(element_count=a.size, struct_size=4)
    b.copy_from_numpy(a)
    bufs.append(b)
tex = dev.create_texture(format=spy.Format.rgba16_float, width=W, height=H)
vars_ = {"gOutput": tex, "gParams": p, "gLabels": bufs[4], "gPreds": bufs[5], **{f"gIntensity{m}": bufs[m] for m in range(4)}}
for _ in range(5):
    k.dispatch(thread_count=[W, H, 1], vars=vars_)
torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
for _ in range(N):
    k.dispatch(thread_count=[W, H, 1], vars=vars_)
t_host = (time.perf_counter() - t0) / N
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / N
print(f"host side of dispatch: {t_host * 1e3:.3f} ms per frame; incl. waiting for the GPU at the end: {t_all * 1e3:.3f} ms per frame")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(N):
    k.dispatch(thread_count=[W, H, 1], vars=vars_)
e1.record(); torch.cuda.synchronize()
print(f"GPU timeline: {e0.elapsed_time(e1) / N:.3f} ms per frame (back-to-back dispatches)")
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    k.dispatch(thread_count=[W, H, 1], vars=vars_)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22)
print(s.getvalue()[:6000])
