#!/usr/bin/env python3
"""BASELINE config 5 frames on one GPU: 256^3 x 4 modalities + seg, 512^2, 256 samples/ray, per-sample MLP.
One line per (network, chunk size): frame ms, queries, live samples, end-to-end TFLOP/s.
    python3 tools/c5_bench.py [--nets siren,fourier] [--chunks 32,64,0] [--alpha 0.4]     (chunk 0 = whole-ray three-pass form)
"""
import argparse
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mrirt  # noqa: E402
from mrirt import inr, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nets", default="siren,fourier")
ap.add_argument("--chunks", default="32,96,0")
ap.add_argument("--layout", default="mod4", choices=["mod4", "quad"], help="one float4 grid of all four modalities, or four quad grids")
ap.add_argument("--alpha", type=float, default=0.4)
ap.add_argument("--frames", type=int, default=5)
a = ap.parse_args()

n, image, steps = 256, 512, 256
vols = [synth.synth_volume(n, 1234 + m, phase=0.3 * m) for m in range(4)]
lab = synth.synth_labels(n)
zmu = [float(v[v != 0].mean()) for v in vols]
zsg = [float(v[v != 0].std() + 1e-6) for v in vols]
p5 = synth.brats_scene(n, image, steps, channels=4, show_seg=True, show_pred=True, intensity_alpha=a.alpha)
gq = [mrirt.upload_grid(v, (n, n, n), "quad") for v in vols]
gm = mrirt.upload_mod4(vols, (n, n, n))
gl = mrirt.upload_grid(lab, (n, n, n), "brick")
rng = np.random.default_rng(0)
for name in a.nets.split(","):
    if name == "siren":
        dims = [7, 256, 256, 256, 256, 4]
        params = [{"W": rng.uniform(-1, 1, (dims[i], dims[i + 1])).astype(np.float32) * np.float32(math.sqrt(6.0 / dims[i]) / (30.0 if i == 0 else 1.0)),
                   "b": rng.uniform(-0.05, 0.05, dims[i + 1]).astype(np.float32)} for i in range(5)]
        net = inr.pack_mlp(params, inr.KIND_SIREN, 0, 4)
    else:
        K = 16
        dims = [3 + 6 * K + 4] + [256] * 4 + [4]
        params = [{"W": (rng.uniform(-1, 1, (dims[i], dims[i + 1])) * math.sqrt(6 / (dims[i] + dims[i + 1]))).astype(np.float32),
                   "b": np.zeros(dims[i + 1], np.float32)} for i in range(5)]
        net = inr.pack_mlp(params, inr.KIND_FOURIER_RELU, K, 4)
    flop = 2 * sum(dims[i] * dims[i + 1] for i in range(5))
    for chunk in [int(c) for c in a.chunks.split(",")]:
        kw = dict(one_pass=True) if chunk == 0 else dict(chunk_steps=chunk)
        gv = gq if (chunk == 0 or a.layout == "quad") else gm            # (the whole-ray form marches with the K1 kernels: quad grids)
        _, aux = inr.render_brats_inr(p5, gv, net, zmu, zsg, labels=gl, return_aux=True, **kw)
        ts = []
        for _ in range(a.frames):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); inr.render_brats_inr(p5, gv, net, zmu, zsg, labels=gl, **kw); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ms = float(np.median(ts))
        print(f"C5 {name:8s} {dims[0]}-4x256-4 {'quad' if gv is gq else 'mod4'} chunk {chunk or 'whole-ray':>9}: {ms:7.3f} ms/frame, queries {aux['queries'] / 1e6:6.2f} M, "
              f"live {aux['live_samples'] / 1e6:6.2f} M, {aux['queries'] * flop / ms / 1e9:6.0f} TFLOP/s issued, "
              f"{aux['live_samples'] * flop / ms / 1e9:6.0f} TFLOP/s useful", flush=True)
