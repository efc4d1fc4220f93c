#!/usr/bin/env python3
"""Near-tie refinement, measured: per network — calibration record, fraction of points marked, argmax agreement with
an fp64 evaluation before / after refinement, error of the split-bf16 logits, and the cost of the extra pass.
    python3 tools/inr_refine_check.py [n_points] > profiles/r03_inr_refine.txt
The mark width and the no-refinement switch travel in MrirtInrDesc.tieSigmas / .flags (inr.with_flags)."""
import math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mrirt
from mrirt import inr

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
rng = np.random.default_rng(11)


def fp64_siren(params, x, w0=30.0):
    h = np.sin(w0 * (x @ params[0]["W"].astype(np.float64)) + params[0]["b"])
    for p in params[1:-1]:
        h = np.sin(h @ p["W"].astype(np.float64) + p["b"])
    return h @ params[-1]["W"].astype(np.float64) + params[-1]["b"]


def fp64_relu(params, x):
    h = x
    for p in params[:-1]:
        h = np.maximum(h @ p["W"].astype(np.float64) + p["b"], 0.0)
    return h @ params[-1]["W"].astype(np.float64) + params[-1]["b"]


def fourier64(coords, feats, K):
    c = coords.astype(np.float64)
    ang = c[..., None] * np.arange(1, K + 1)[None, None, :] * np.pi
    ff = np.concatenate([np.sin(ang), np.cos(ang)], -1).reshape(c.shape[0], -1)
    return np.concatenate([c, ff, feats.astype(np.float64)], 1)


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev]))


def study(name, net, coords, feats, want):
    wa = want.argmax(1)
    scale = np.abs(want).max()
    cal = inr.calibration(net)
    c = torch.from_numpy(coords).cuda() if coords is not None else None
    f = torch.from_numpy(feats).cuda()
    n = f.shape[0]
    print(f"== {name}: n={n}, logit range {scale:.3f}; calibration rms {cal['rms_error']:.3e} (= {cal['rms_error'] / scale:.2e} of range), "
          f"max {cal['max_error']:.3e}, max|logit| on the calibration set {cal['max_logit']:.3f}")
    raw = inr.with_flags(net, no_refine=True)
    lg, a0 = inr._forward(raw, c, f, n, True, True)
    a0 = a0.cpu().numpy(); lg = lg.cpu().numpy().astype(np.float64)
    t0 = timed(lambda: inr._forward(raw, c, f, n, False, True))
    srt = np.sort(lg, 1)
    gap = srt[:, -1] - srt[:, -2]
    print(f"   bf16 pass: max|dlogit| {np.abs(lg - want).max() / scale:.2e} of range, rms {np.sqrt(((lg - want) ** 2).mean()) / scale:.2e}; "
          f"agreement {np.mean(a0 == wa):.5f} ({int((a0 != wa).sum())} flips); {t0:.3f} ms")
    lr, ar = inr._forward(net, c, f, n, True, True, refined=True)
    lr = lr.cpu().numpy().astype(np.float64); ar = ar.cpu().numpy()
    tr = timed(lambda: inr._forward(net, c, f, n, False, True, refined=True), reps=1)
    print(f"   split-bf16 pass on every point: max|dlogit| {np.abs(lr - want).max() / scale:.2e} of range; agreement {np.mean(ar == wa):.6f} "
          f"({int((ar != wa).sum())} flips); {tr:.3f} ms")
    for sig in (2.0, 2.5, 3.0, 4.0, 5.0):
        wide = inr.with_flags(net, tie_sigmas=sig)
        _, a1 = inr._forward(wide, c, f, n, False, True)
        a1 = a1.cpu().numpy()
        marked = np.mean(gap < sig * math.sqrt(2) * cal["rms_error"])
        t1 = timed(lambda: inr._forward(wide, c, f, n, False, True))
        assert (a1 & 0x4000).sum() == 0
        print(f"   sigmas {sig:3.1f}: marked {marked:.4f} of the points; agreement {np.mean(a1 == wa):.6f} ({int((a1 != wa).sum())} flips); "
              f"bf16 + refinement {t1:.3f} ms (+{(t1 - t0) / t0 * 100:.1f} %)")


coords = (rng.random((N, 3)) * 2 - 1).astype(np.float32)
feats = rng.standard_normal((N, 4)).astype(np.float32)
for hid_layers in (4, 3):
    dims = [7] + [256] * hid_layers + [4]
    params = [{"W": rng.uniform(-1, 1, (dims[i], dims[i + 1])).astype(np.float32) * np.float32(math.sqrt(6.0 / dims[i]) / (30.0 if i == 0 else 1.0)),
               "b": rng.uniform(-0.05, 0.05, dims[i + 1]).astype(np.float32)} for i in range(len(dims) - 1)]
    want = fp64_siren(params, np.concatenate([coords, feats], 1).astype(np.float64))
    study(f"SIREN 7-{hid_layers}x256-4 (notebook init)", inr.pack_mlp(params, inr.KIND_SIREN, 0, 4), coords, feats, want)
for K, hidden in ((4, 64), (16, 256), (10, 128)):
    dims = [3 + 6 * K + 4] + [hidden] * 4 + [4]
    params = [{"W": rng.uniform(-1, 1, (dims[i], dims[i + 1])).astype(np.float32) * np.float32(math.sqrt(6 / (dims[i] + dims[i + 1]))),
               "b": rng.uniform(-0.1, 0.1, dims[i + 1]).astype(np.float32)} for i in range(5)]
    want = fp64_relu(params, fourier64(coords, feats, K))
    study(f"Fourier/ReLU K={K} 4x{hidden}", inr.pack_mlp(params, inr.KIND_FOURIER_RELU, K, 4), coords, feats, want)
