#!/usr/bin/env python3
"""The weight-stationary INR kernel against the streaming kernel on the same packed image: logits and classes must be the same
bits (the tests' check, runnable against a variant library: MRIRT_LIB=build_exp/libmrirt_<name>.so), then a timing of the bf16 pass."""
import math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrirt import inr
rng = np.random.default_rng(77)
dims = [7, 256, 256, 256, 256, 4]
params = [{"W": (rng.uniform(-1, 1, (dims[i], dims[i + 1])) * math.sqrt(6.0 / dims[i]) / (30.0 if i == 0 else 1.0)).astype(np.float32),
           "b": rng.uniform(-0.05, 0.05, dims[i + 1]).astype(np.float32)} for i in range(5)]
net = inr.pack_mlp(params, inr.KIND_SIREN, 0, 4)
ok = True
for n in (96 * 256 * 2 + 61, 5000, 97, 96 * 256 * 7):
    coords = (torch.rand((n, 3), device="cuda") * 2 - 1).contiguous()
    feats = torch.randn((n, 4), device="cuda").contiguous()
    lw, cw = inr._forward(net, coords, feats, n, True, True)
    ls, cs = inr._forward(inr.with_flags(net, no_weight_stationary=True), coords, feats, n, True, True)
    same = bool(torch.equal(lw, ls) and torch.equal(cw, cs))
    ok = ok and same
    print(f"n = {n}: weight-stationary == streaming: {same}" + ("" if same else f"  (max |dlogit| {float((lw - ls).abs().max()):.3e}, classes differ on {int((cw != cs).sum())})"))
nq = 512 * 512 * 256
coords = torch.rand((nq, 3), device="cuda") * 2 - 1
feats = torch.randn((nq, 4), device="cuda")
raw = inr.with_flags(net, mark_only=True)
for _ in range(3):
    inr._forward(raw, coords, feats, nq, False, True)
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); inr._forward(raw, coords, feats, nq, False, True); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ms = float(np.mean(ts))
print(f"bf16 pass, 67.1 M queries: {ms:.3f} ms = {398848 * nq / ms / 1e9:.1f} TFLOP/s = {398848 * nq / ms / 1e9 / 2500:.4f} of 2.5 PF; all equal: {ok}")
