#!/usr/bin/env python3
"""Shader clock and package power while one kernel runs back to back (rocm-smi polled from a thread): is a kernel's clock
what the power cap leaves it?    python3 tools/power_clocks.py            (writes a table to stdout)
Workloads: the weight-stationary INR kernel, the near-tie refinement alone, the C3 march, the C2 march (MOD4), idle."""
import json, math, os, subprocess, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mrirt
from mrirt import inr, synth


def poll(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)
            c = d.get("card0", next(iter(d.values())))
            out.append({k: v for k, v in c.items() if "sclk" in k.lower() or "power" in k.lower() or "mclk" in k.lower()})
        except Exception as e:          # noqa: BLE001 - telemetry is best effort
            out.append({"error": repr(e)[:120]})
        time.sleep(0.15)


def measure(name, fn, seconds=4.0):
    fn(); torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=poll, args=(stop, samples)); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            fn()
        torch.cuda.synchronize(); n += 8
    dt = (time.perf_counter() - t0) / n
    stop.set(); th.join()
    good = [s for s in samples[2:] if "error" not in s]          # (the first samples straddle the ramp)
    print(f"== {name}: {dt * 1e3:.3f} ms per launch, {len(good)} telemetry samples")
    keys = sorted({k for s in good for k in s})
    for k in keys:
        vals = []
        for s in good:
            v = str(s.get(k, "")).replace("(", "").replace(")", "").replace("Mhz", "").replace("MHz", "")
            try:
                vals.append(float(v))
            except ValueError:
                pass
        if vals:
            print(f"   {k:48s} median {np.median(vals):9.1f}   min {min(vals):9.1f}   max {max(vals):9.1f}")
    if not good:
        print("   no telemetry:", samples[:2])


rng = np.random.default_rng(0)
dims = [7, 256, 256, 256, 256, 4]
params = [{"W": (rng.uniform(-1, 1, (dims[i], dims[i + 1])) * math.sqrt(6.0 / dims[i]) / (30.0 if i == 0 else 1.0)).astype(np.float32),
           "b": rng.uniform(-0.05, 0.05, dims[i + 1]).astype(np.float32)} for i in range(5)]
net = inr.pack_mlp(params, inr.KIND_SIREN, 0, 4)
nq = 1 << 24
c = torch.rand((nq, 3), device="cuda") * 2 - 1
f = torch.randn((nq, 4), device="cuda")
mark_only = inr.with_flags(net, mark_only=True)
measure("idle (host sleeps)", lambda: time.sleep(0.05), seconds=2.0)
measure("INR weight-stationary kernel, 16.8 M queries", lambda: inr._forward(mark_only, c, f, nq, False, True))
nr = 1 << 21
measure("INR near-tie refinement alone (every point), 2.1 M queries", lambda: inr._forward(net, c[:nr], f[:nr], nr, False, True, refined=True))
n = 512
vol = synth.synth_volume(n)
g = mrirt.upload_grid(vol, (n, n, n), "vga")
p3 = synth.brats_scene(n, 1024, 512, channels=1, intensity_alpha=16.0)
ext = dict(synth.SHADE_EXT, layout="vga")
out = torch.empty((1024, 1024, 4), device="cuda")
measure("C3 march (512^3, 1024^2, 512 steps, shaded, VGA)", lambda: mrirt.render_brats(p3, [g], out=out, ext=ext))
measure("C3 march, FAST math (no fp64 exp, no IEEE divisions)", lambda: mrirt.render_brats(p3, [g], out=out, ext=dict(ext, math="fast")))
del g
n = 256
g = mrirt.upload_grid(synth.synth_volume(n), (n, n, n), "vga")
p3s = synth.brats_scene(n, 1024, 512, channels=1, intensity_alpha=16.0)
measure("C3 march on a 256^3 volume (Infinity-Cache resident), same image and steps", lambda: mrirt.render_brats(p3s, [g], out=out, ext=ext))
del g
n = 256
vols = [synth.synth_volume(n, 1234 + m, phase=0.3 * m) for m in range(4)]
g4 = mrirt.upload_mod4(vols, (n, n, n))
gl = mrirt.upload_label_cells(synth.synth_labels(n), None, (n, n, n))
p2 = synth.brats_scene(n, 512, 256, channels=4, show_seg=True, intensity_alpha=0.4)
out2 = torch.empty((512, 512, 4), device="cuda")
measure("C2 march (256^3 x 4, 512^2, 256 steps, MOD4 + label cells)", lambda: mrirt.render_brats(p2, [g4] * 4, labels=gl, out=out2, ext=dict(layout="mod4")))
