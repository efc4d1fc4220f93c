#!/usr/bin/env python3
"""inr_refine_kernel alone: every point of a resident batch through the split-bf16 pass (mrirt_inr_forward_refined), so that
the time per 128-point batch per CU can be read off directly.
    python3 tools/refine_bench.py [n_points] [net=siren|fourier]
Library variants: MRIRT_LIB=build_exp/libmrirt_<name>.so (tools/build_variant.sh)."""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mrirt  # noqa: E402,F401
from mrirt import inr  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_097_152
kind = sys.argv[2] if len(sys.argv) > 2 else "siren"
rng = np.random.default_rng(0)
torch.manual_seed(0)
if kind == "siren":
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
c = torch.rand(N, 3, device="cuda") * 2 - 1
f = torch.randn(N, 4, device="cuda")
flop = 2 * sum(dims[i] * dims[i + 1] for i in range(5))


def run():
    return inr._forward(net, c, f, N, False, True, refined=True)[1]


a0 = run(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); a1 = run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ms = float(np.median(ts))
cus = torch.cuda.get_device_properties(0).multi_processor_count
per_batch = ms * 1e3 / (N / 128 / cus)
print(f"refine {kind} n={N}: {ms:.3f} ms/launch, {per_batch:.2f} us per 128-point batch per CU ({per_batch / 26:.2f} us per out tile), "
      f"{3 * N * flop / ms / 1e9:.0f} TFLOP/s issued (3 MFMAs per product), checksum {int(a1.to(torch.int64).sum())}", flush=True)
