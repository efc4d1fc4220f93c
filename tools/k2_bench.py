#!/usr/bin/env python3
"""K2 (volume_cs) at the reference's asset size: 180x216x180 u8, fovY 72deg, radius 4.2 (app.py:34,336)."""
import os, sys, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mrirt
from mrirt import synth, volume
from oracle import oracle_c
dims = (180, 216, 180)
f = synth.synth_volume(0, 1234, dims=dims)
u8 = np.rint(f * 255).astype(np.uint8)
pk = volume.pack_u8_as_u32x4(u8)
cam = synth.bench_camera(radius=4.2, phi_deg=80, theta_deg=25)
def timeit(fn, rounds=9):
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
print(f"{'image':>6} {'steps':>5} {'near':>4} {'far':>4} {'mode':>6} {'math':>6} {'ms':>8} {'live Ms':>9} {'Gsamp/s':>8} {'nominal Gs/s':>12} {'err':>9}")
modes = [m for m in os.environ.get("K2_MODES", "u32x4,u8,cell8,f32").split(",")]
for image, steps, near, far in ((1024, 512, 1.5, 4.5), (1024, 64, 4.3, 4.4), (2048, 512, 1.5, 6.9)):
    p = synth.volume_scene(0, image, steps, near, far, fov_deg=72.0, camera=cam, dims=dims)
    ref = None
    if image == 1024 and steps == 64:
        ref = oracle_c.volume_cs(p, pk, mode="u32x4")
    for mode in modes:
        v = {"u32x4": torch.from_numpy(pk.view(np.int32)).cuda(), "u8": torch.from_numpy(u8).cuda(), "f32": torch.from_numpy(f).cuda()}.get(mode)
        if v is None:
            v = mrirt.render.build_cell8(u8, dims, "u8")
        for math_ in ("strict", "fast"):
            e = dict(math=math_)
            img, st = mrirt.render_volume_u8(p, v, mode=mode, ext=e, stats=True)
            err = float(np.abs(img.cpu().numpy() - ref).max()) if ref is not None and mode != "f32" else float("nan")
            ms = timeit(lambda: mrirt.render_volume_u8(p, v, mode=mode, ext=e))
            print(f"{image:6d} {steps:5d} {near:4.1f} {far:4.1f} {mode:>6} {math_:>6} {ms:8.3f} {st['live_samples']/1e6:9.2f} {st['live_samples']/ms/1e6:8.2f} {image*image*steps/ms/1e6:12.2f} {err:9.2e}")
