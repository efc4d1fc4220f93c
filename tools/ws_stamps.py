#!/usr/bin/env python3
"""Decode the s_memtime stamps of a -DMRIRT_WS_STAMPS build (MRIRT_LIB=build_exp/libmrirt_WSSTAMPS.so):
median cycles per phase of one steady-state round, over all workgroups, per wave."""
import os, sys, math, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import mrirt
from mrirt import inr
rng = np.random.default_rng(0)
dims = [7, 256, 256, 256, 256, 4]
params = [{"W": (rng.uniform(-1, 1, (dims[i], dims[i + 1])) * math.sqrt(6.0 / dims[i]) / (30.0 if i == 0 else 1.0)).astype(np.float32),
           "b": rng.uniform(-0.05, 0.05, dims[i + 1]).astype(np.float32)} for i in range(5)]
net = inr.with_flags(inr.pack_mlp(params, inr.KIND_SIREN, 0, 4), no_refine=True)      # the second pass would overwrite the stamps
n = 256 * 96 * 40
coords = torch.rand((n, 3), device="cuda") * 2 - 1
feats = torch.randn((n, 4), device="cuda")
for _ in range(3):
    logits, cls = inr._forward(net, coords, feats, n, True, True)
torch.cuda.synchronize()
st = logits.view(torch.int64).reshape(-1)[:256 * 4 * 24].cpu().numpy().reshape(256, 4, 24)
names = ["L0+head", "barrier", "hidden1", "barrier", "hidden2", "barrier", "hidden3", "stage+barrier"]
d = np.diff(st[:, :, :8], axis=2)            # [block][wave][7]
labels = ["L0+head", "bar", "hidden1(+bar)", "hidden2", "bar", "hidden3", "stage+bar"]
for wv in range(4):
    print("wave", wv, " ".join(f"{labels[k]}={int(np.median(d[:, wv, k]))}" for k in range(7)), "round", int(np.median(st[:, wv, 7] - st[:, wv, 0])))
if True:
  for wv in (0, 3):
    print("wave", wv, "hidden1 passes:", "prologue", int(np.median(st[:, wv, 8] - st[:, wv, 2])),
          [int(np.median(st[:, wv, 9 + k] - st[:, wv, 8 + k])) for k in range(6)], "tail", int(np.median(st[:, wv, 3] - st[:, wv, 14])))
for wv in (0, 3):
    print("wave", wv, "L0+head passes:", "prologue", int(np.median(st[:, wv, 15] - st[:, wv, 0])),
          [int(np.median(st[:, wv, 16 + k] - st[:, wv, 15 + k])) for k in range(6)], "tail (last activation, head end, stores)", int(np.median(st[:, wv, 1] - st[:, wv, 21])))
if False:
  for wv in range(4):
    print("v2 wave", wv, "L0+head", int(np.median(st[:, wv, 1] - st[:, wv, 0])), "H1", int(np.median(st[:, wv, 2] - st[:, wv, 1])),
          "H2", int(np.median(st[:, wv, 3] - st[:, wv, 2])), "H3", int(np.median(st[:, wv, 4] - st[:, wv, 3])), "round", int(np.median(st[:, wv, 4] - st[:, wv, 0])))
  # clock: stamps are shader cycles; wall time of the launch
  import time
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(5): inr._forward(net, coords, feats, n, True, True)
  torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
  rounds_per_cu = n / 96 / 256
  print("launch", dt * 1e3, "ms; rounds per CU", rounds_per_cu, "=> wall ns per round", dt / rounds_per_cu * 1e9)

import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): inr._forward(net, coords, feats, n, True, True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
rounds_per_cu = n / 96 / 256
rc = float(np.median(st[:, 0, 7] - st[:, 0, 0]))
print("launch %.3f ms; %.1f rounds per CU => %.0f ns per round; %.0f cycles per round => %.2f GHz" % (dt * 1e3, rounds_per_cu, dt / rounds_per_cu * 1e9, rc, rc / (dt / rounds_per_cu * 1e9)))
