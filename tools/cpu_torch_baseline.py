#!/usr/bin/env python3
"""The optional "PyTorch path" CPU number of BASELINE.md section 3 (CPU-torch): K1 without overlays, marched
with ``torch.nn.functional.grid_sample`` (5-D input = trilinear, ``align_corners=True``, border padding) on the
host cores.  A reported baseline only — it is NOT bit-faithful to the shader (grid_sample fuses and reorders the
interpolation) and is not used for parity; the script prints how far it lands from the oracle.

    python tools/cpu_torch_baseline.py [volume_n image steps]
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load_synth():
    import mrirt                   # importing the package does not load libmrirt.so; synth is NumPy only
    return mrirt.synth


def render(p, vol, n):
    W, H = int(p["imageSize"][0]), int(p["imageSize"][1])
    eye, U, V, Wv = (torch.tensor(np.asarray(p[k], np.float32)) for k in ("eye", "U", "V", "W"))
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    uvx = ((xs + 0.5) / W) * 2 - 1
    uvy = ((ys + 0.5) / H) * 2 - 1
    f = 1.0 / np.tan(0.5 * float(p["fovY"]))
    aspect = W / max(1, H)
    d = torch.stack([uvx * aspect / f, -uvy / f, torch.ones_like(uvx)], -1)
    d = d / d.norm(dim=-1, keepdim=True)
    rd = d[..., 0:1] * U + d[..., 1:2] * V + d[..., 2:3] * Wv
    rd = rd / rd.norm(dim=-1, keepdim=True)
    bmin = torch.tensor(np.asarray(p["volMin"], np.float32))
    vox = torch.tensor(np.asarray(p["voxelSize"], np.float32))
    dims = torch.tensor([float(v) for v in p["dims"]])
    bmax = bmin + vox * dims
    safe = torch.where(rd.abs() < 1e-6, torch.full_like(rd, 1e-6), rd)
    ta, tb = (bmin - eye) / safe, (bmax - eye) / safe
    tmin = torch.minimum(ta, tb).amax(-1)
    tmax = torch.maximum(ta, tb).amin(-1)
    hit = tmax >= tmin.clamp_min(0)
    t0 = tmin.clamp_min(max(0.0, float(p["nearT"])))
    t1 = tmax
    dt = float(p["stepSize"])
    grid5 = torch.from_numpy(vol.reshape(n, n, n))[None, None]            # (1,1,Z,Y,X)
    C = torch.zeros(H, W)
    T = torch.ones(H, W)
    t = t0.clone()
    tf_lo, ww, a = float(p["wl"]) - 0.5 * float(p["ww"]), float(p["ww"]), float(p["intensityAlpha"])
    live = 0
    steps = int(np.ceil(float((t1 - t0)[hit].max()) / dt)) + 1
    for _ in range(steps):
        act = hit & (t < t1) & (T > 0.01)
        if not bool(act.any()):
            break
        live += int(act.sum())
        pos = eye + t[..., None] * rd
        q = ((pos - bmin) / vox).clamp(min=torch.zeros(3), max=dims - 1.001)
        g = 2 * q / (dims - 1) - 1                                         # x,y,z in [-1,1] (align_corners)
        s = F.grid_sample(grid5, g[None, None], mode="bilinear", padding_mode="border", align_corners=True)[0, 0, 0]
        val = ((s - tf_lo) / ww).clamp(0, 1)
        alpha = torch.where(act & (val > 0), 1 - torch.exp(-val * a * dt), torch.zeros_like(val))
        C = C + alpha * T * val
        T = T * (1 - alpha)
        t = torch.where(act, t + dt, t)
    return C, live


def main():
    n, image, steps = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (128, 256, 128)
    synth = load_synth()
    vol = synth.synth_volume(n)
    p = synth.brats_scene(n, image, steps, channels=1, intensity_alpha=16.0)
    torch.set_grad_enabled(False)
    render(p, vol, n)                                                      # warm-up
    t = time.perf_counter()
    img, live = render(p, vol, n)
    dt = time.perf_counter() - t
    line = f"CPU-torch (grid_sample, {torch.get_num_threads()} threads): {n}^3, {image}^2, {steps} steps: {dt * 1e3:.0f} ms, {live / dt / 1e6:.1f} Msamples/s ({live} live samples)"
    try:
        from oracle import oracle_c
        ref = oracle_c.brats_main(p, [vol], None, None)[..., 0]
        line += f"; max |image - oracle| = {float(np.abs(img.numpy() - ref).max()):.2e} (not a parity path)"
    except Exception as e:                                                 # the oracle library is optional here
        line += f"; oracle comparison skipped ({type(e).__name__})"
    print(line)


if __name__ == "__main__":
    main()
