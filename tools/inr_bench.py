#!/usr/bin/env python3
"""Throughput of the fused MLP kernel: queries/s and bf16 TFLOP/s (2*sum(in*out) flop per query)."""
import os, sys, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mrirt
from mrirt import inr
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512 * 512 * 256
rng = np.random.default_rng(0)
ZERO = os.environ.get("INR_ZERO") == "1"      # DVFS probe: the same instruction stream on all-zero operands
def net(dims):
    if ZERO:
        return [{"W": np.zeros((dims[i], dims[i+1]), np.float32), "b": np.zeros(dims[i+1], np.float32)} for i in range(len(dims) - 1)]
    return [{"W": (rng.uniform(-1, 1, (dims[i], dims[i+1])) * math.sqrt(6 / (dims[i] + dims[i+1]))).astype(np.float32),
             "b": np.zeros(dims[i+1], np.float32)} for i in range(len(dims) - 1)]
cases = {"siren 7-4x256-4": ([7] + [256] * 4 + [4], inr.KIND_SIREN, 0, 4),
         "fourier 103-4x256-4": ([103] + [256] * 4 + [4], inr.KIND_FOURIER_RELU, 16, 4),
         "fourier 31-4x64-4": ([31] + [64] * 4 + [4], inr.KIND_FOURIER_RELU, 4, 4)}
coords = torch.rand((n, 3), device="cuda") * 2 - 1
feats = torch.rand((n, 4), device="cuda")
if ZERO:
    coords.zero_(); feats.zero_()
out = torch.empty(n, dtype=torch.int16, device="cuda")
import ctypes as C
from mrirt import _lib
for name, (dims, kind, K, M) in cases.items():
    p = net(dims)
    pk = inr.pack_mlp(p, kind, K, M)
    flop = 2 * sum(dims[i] * dims[i+1] for i in range(len(dims) - 1))
    ts = []
    for it in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = _lib.lib().mrirt_inr_forward(C.byref(pk.desc), C.c_void_p(coords.data_ptr()), C.c_void_p(feats.data_ptr()), n,
                                          None, C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        e1.record(); torch.cuda.synchronize(); assert rc == 0
        ts.append(e0.elapsed_time(e1))
    ms = min(ts[1:])
    print(f"{name:24s} n={n} {ms:9.3f} ms  {n / ms / 1e3:9.1f} Mquery/s  {flop * n / ms / 1e9:8.1f} TFLOP/s  ({flop} flop/query)")
