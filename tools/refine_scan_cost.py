"""What the refinement launch costs BEFORE it evaluates a single point: scanning the class array for marks.
    python3 tools/refine_scan_cost.py"""
import math, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, mrirt
from mrirt import inr
rng = np.random.default_rng(0)
dims = [7, 256, 256, 256, 256, 4]
params = [{"W": (rng.uniform(-1, 1, (dims[i], dims[i + 1])) * math.sqrt(6.0 / dims[i]) / (30.0 if i == 0 else 1.0)).astype(np.float32),
           "b": rng.uniform(-0.05, 0.05, dims[i + 1]).astype(np.float32)} for i in range(5)]
net = inr.pack_mlp(params, inr.KIND_SIREN, 0, 4)
for n in (8_600_000, 67_108_864):
    c = torch.rand((n, 3), device="cuda") * 2 - 1
    f = torch.randn((n, 4), device="cuda")
    def timed(nn, reps=5):
        for _ in range(2): inr._forward(nn, c, f, n, False, True)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record(); inr._forward(nn, c, f, n, False, True); b.record()
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in ev]))
    t_mark = timed(inr.with_flags(net, mark_only=True))
    t_scan = timed(inr.with_flags(net, tie_sigmas=1e-6))
    t_split = timed(net)
    print(f"n={n}: main+mark {t_mark:.3f} ms; + a scan-only refinement launch (mark width 1e-6 sigma: nothing to redo) {t_scan - t_mark:+.3f}; "
          f"+ the refinement as shipped {t_split - t_mark:+.3f}")
    del c, f
