import os, sys, math, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import mrirt
from mrirt import inr
rng = np.random.default_rng(23)
dims, w0 = [7, 256, 256, 256, 256, 4], 30.0
params = []
for i in range(len(dims) - 1):
    r = math.sqrt(6.0 / dims[i]) / (w0 if i == 0 else 1.0)
    params.append({"W": rng.uniform(-r, r, (dims[i], dims[i + 1])).astype(np.float32), "b": rng.uniform(-0.05, 0.05, dims[i + 1]).astype(np.float32)})
net = inr.pack_mlp(params, inr.KIND_SIREN, 0, 4, w0=w0)
for n in (96, 192, 1000):
    coords = (rng.random((n, 3)) * 2 - 1).astype(np.float32)
    feats = rng.standard_normal((n, 4)).astype(np.float32)
    c, f = torch.from_numpy(coords).cuda(), torch.from_numpy(feats).cuda()
    got, _ = inr._forward(net, c, f, n, True, True)
    os.environ["MRIRT_INR_NO_WS"] = "1"
    ref, _ = inr._forward(net, c, f, n, True, True)
    os.environ.pop("MRIRT_INR_NO_WS", None)
    d = (got - ref).abs().max(dim=1).values.cpu().numpy()
    bad = np.nonzero(d > 1e-3)[0]
    print(n, "bad points", bad.tolist()[:40], "errs", np.round(d[bad][:10], 3).tolist())
    got2, _ = inr._forward(net, c, f, n, True, True)
    print("   deterministic:", bool(torch.equal(got, got2)))
