#!/usr/bin/env python3
"""Decode the s_memtime stamps of a -DMRIRT_REF_STAMPS build of inr_refine_kernel (bash tools/build_variant.sh REFSTAMPS inr_mlp.hip -DMRIRT_REF_STAMPS; MRIRT_LIB=build_exp/libmrirt_REFSTAMPS.so):
median shader cycles per phase of the second batch of every workgroup, and the clock they imply."""
import math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mrirt  # noqa
from mrirt import inr
rng = np.random.default_rng(0)
dims = [7, 256, 256, 256, 256, 4]
params = [{"W": (rng.uniform(-1, 1, (dims[i], dims[i + 1])) * math.sqrt(6.0 / dims[i]) / (30.0 if i == 0 else 1.0)).astype(np.float32),
           "b": rng.uniform(-0.05, 0.05, dims[i + 1]).astype(np.float32)} for i in range(5)]
net = inr.pack_mlp(params, inr.KIND_SIREN, 0, 4)
n = 256 * 128 * 64
c = torch.rand((n, 3), device="cuda") * 2 - 1
f = torch.randn((n, 4), device="cuda")
for _ in range(3):
    logits, cls = inr._forward(net, c, f, n, True, True, refined=True)
torch.cuda.synchronize()
st = logits.view(torch.int64).reshape(-1)[:256 * 4 * 64].cpu().numpy().reshape(256, 4, 64)
med = lambda x: int(np.median(x))
for w in range(4):
    print(f"wave {w}: L0 {med(st[:, w, 1] - st[:, w, 0])}  hidden1 {med(st[:, w, 2] - st[:, w, 1])}  hidden2 {med(st[:, w, 3] - st[:, w, 2])}  "
          f"hidden3 {med(st[:, w, 4] - st[:, w, 3])}  head {med(st[:, w, 5] - st[:, w, 4])}  batch {med(st[:, w, 5] - st[:, w, 0])}")
for w in (0, 3):
    print(f"wave {w} hidden2 tiles [issue+MFMAs, tile_done wait+barrier]:",
          [(med(st[:, w, 9 + 3 * o] - st[:, w, 8 + 3 * o]), med(st[:, w, 10 + 3 * o] - st[:, w, 9 + 3 * o])) for o in range(8)],
          "tail activation", med(st[:, w, 3] - st[:, w, 10 + 21]))
for w in (0, 3):
    print(f"wave {w} hidden2 tile 2: issue_tile {med(st[:, w, 40] - st[:, w, 14])}, to first k step issued {med(st[:, w, 41] - st[:, w, 40])}, k steps",
          [med(st[:, w, 42 + q] - st[:, w, 41 + q]) for q in range(15)])
    print(f"wave {w} head: before issue (tail activation) {med(st[:, w, 32] - st[:, w, 4])}, issue_tile {med(st[:, w, 33] - st[:, w, 32])}, bias + MFMAs {med(st[:, w, 34] - st[:, w, 33])}, "
          f"top-2 {med(st[:, w, 35] - st[:, w, 34])}, tile_done {med(st[:, w, 5] - st[:, w, 35])}")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): inr._forward(net, c, f, n, True, True, refined=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
per_cu = n / 128 / 256
cyc = float(np.median(st[:, 0, 5] - st[:, 0, 0]))
full = float(np.median(st[:, 0, 6] - st[:, 0, 0]))
print("launch %.3f ms; %.1f batches per CU => %.0f ns per batch; %.0f cycles L0..head, %.0f cycles batch start to batch start => %.2f GHz" % (dt * 1e3, per_cu, dt / per_cu * 1e9, cyc, full, full / (dt / per_cu * 1e9)))
