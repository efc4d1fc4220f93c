#!/usr/bin/env python3
"""BASELINE configs C1 (CPU oracle, K2 fp32 ortho), C2 (K1 reference semantics: 4 modalities + seg
overlay, 256^3, 512^2, 256 steps) and C3 on one GPU; prints one row per run."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mrirt
from mrirt import synth
from oracle import oracle_c, oracle_np

def timeit(fn, rounds=7):
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))

rows = []
# ---- C1: 128^3 fp32, 256x256 orthographic, 64 steps: K2 loop, CPU oracle (plumbing config) + GPU beside it
f = synth.synth_volume(128)
p = synth.volume_scene(128, 256, 64)
ext = dict(cameraMode=1, orthoHalfHeight=1.1)
t = time.perf_counter(); ref, aux = oracle_c.volume_cs(p, f, mode="f32", ext=ext, return_aux=True); dt_c = time.perf_counter() - t
t = time.perf_counter(); refn = oracle_np.volume_cs(p, f, mode="f32", ext=ext); dt_np = time.perf_counter() - t
assert np.array_equal(ref, refn)
g = torch.from_numpy(f).cuda()
img, st = mrirt.render_volume_u8(p, g, mode="f32", ext=ext, stats=True)
ms = timeit(lambda: mrirt.render_volume_u8(p, g, mode="f32", ext=ext))
rows.append(("C1 128^3 256^2 64 ortho K2-f32", f"CPU-np {dt_np*1e3:.0f} ms ({aux['live_samples']/dt_np/1e6:.1f} Ms/s 1 thr), CPU-omp {dt_c*1e3:.1f} ms ({aux['live_samples']/dt_c/1e6:.0f} Ms/s {os.cpu_count()} thr)",
             ms, aux["live_samples"], float(np.abs(img.cpu().numpy() - ref).max())))
# ---- C2: 256^3 BraTS-shaped (4 channels + labels), 512^2 perspective, 256 steps
n = 256
vols = [synth.synth_volume(n, 1234 + m, phase=0.3 * m) for m in range(4)]
lab = synth.synth_labels(n)
p2 = synth.brats_scene(n, 512, 256, channels=4, show_seg=True, intensity_alpha=0.4)
t = time.perf_counter(); ref2, aux2 = oracle_c.brats_main(p2, vols, lab, None, return_aux=True); dt2 = time.perf_counter() - t
for layout in ("linear", "brick", "quad", "quad+cells", "mod4", "mod4+cells"):
    cells = layout.endswith("+cells")               # the seg overlay as label cells (what the shim binds on QUAD / MOD4 frames)
    base = layout.split("+")[0]
    gv = [mrirt.upload_mod4(vols, (n, n, n))] * 4 if base == "mod4" else [mrirt.upload_grid(v, (n, n, n), base) for v in vols]
    gl = mrirt.upload_label_cells(lab, None, (n, n, n)) if cells else mrirt.upload_grid(lab, (n, n, n), "linear" if layout == "linear" else "brick")
    for math_ in ("strict", "fast"):
        e = dict(math=math_)
        img = mrirt.render_brats(p2, gv, gl, ext=e)
        err = float(np.abs(img.cpu().numpy() - ref2).max())
        ms = timeit(lambda: mrirt.render_brats(p2, gv, gl, ext=e))
        rows.append((f"C2 256^3x4ch+seg 512^2 256 {layout} {math_}", f"CPU-omp {dt2*1e3:.0f} ms ({aux2['live_samples']/dt2/1e6:.0f} Ms/s)", ms, aux2["live_samples"], err))
    del gv
# single-channel variant of C2 for the GB/s headline
p2s = synth.brats_scene(n, 512, 256, channels=1, intensity_alpha=0.4)
gq = mrirt.upload_grid(vols[0], (n, n, n), "quad")
_, sts = mrirt.render_brats(p2s, [gq], stats=True)
ms = timeit(lambda: mrirt.render_brats(p2s, [gq]))
rows.append(("C2-1ch 256^3 512^2 256 quad strict", "", ms, sts["live_samples"], 0.0))
print(f"{'config':48s} {'GPU ms':>8s} {'live Ms':>9s} {'Gsamp/s':>8s} {'max|err|':>9s}  cpu")
for name, cpu, ms, live, err in rows:
    print(f"{name:48s} {ms:8.3f} {live/1e6:9.2f} {live/ms/1e6:8.2f} {err:9.2e}  {cpu}")

# ---- C5: per-sample INR query, 512^2 x 256 steps on the 256^3 4-modality scene, Fourier/ReLU 103 -> 4x256 -> 4
from mrirt import inr
import math
rng = np.random.default_rng(0)
K = 16
sizes = [3 + 6 * K + 4] + [256] * 4 + [4]
mlp = [{"W": (rng.uniform(-1, 1, (sizes[i], sizes[i+1])) * math.sqrt(6 / (sizes[i] + sizes[i+1]))).astype(np.float32),
        "b": np.zeros(sizes[i+1], np.float32)} for i in range(5)]
net = inr.pack_mlp(mlp, inr.KIND_FOURIER_RELU, K, 4)
zmu = [float(v[v != 0].mean()) for v in vols]
zsg = [float(v[v != 0].std() + 1e-6) for v in vols]
p5 = synth.brats_scene(n, 512, 256, channels=4, show_seg=True, show_pred=True, intensity_alpha=0.4)
gv = mrirt.upload_mod4(vols, (n, n, n))
gl = mrirt.upload_grid(lab, (n, n, n), "brick")
img, aux = inr.render_brats_inr(p5, gv, net, zmu, zsg, labels=gl, return_aux=True)
ms5 = timeit(lambda: inr.render_brats_inr(p5, gv, net, zmu, zsg, labels=gl), rounds=5)
flop = 2 * sum(sizes[i] * sizes[i+1] for i in range(5))
print(f"C5 256^3x4ch 512^2 256 steps, Fourier 103-4x256-4: {ms5:.2f} ms/frame, {aux['queries']/1e6:.1f} M MLP queries (live samples, ERT-aware passes), "
      f"{aux['queries']/ms5/1e3:.0f} Mquery/s end to end, {flop*aux['queries']/ms5/1e9:.0f} TFLOP/s end to end")
