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
for n in (1, 95, 96, 97, 4096, 300000):
    coords = (rng.random((n, 3)) * 2 - 1).astype(np.float32)
    feats = rng.standard_normal((n, 4)).astype(np.float32)
    x = np.concatenate([coords, feats], axis=1).astype(np.float64)
    h = np.sin(w0 * (x @ params[0]["W"].astype(np.float64)) + params[0]["b"])
    for p in params[1:-1]:
        h = np.sin(h @ p["W"].astype(np.float64) + p["b"])
    want = h @ params[-1]["W"].astype(np.float64) + params[-1]["b"]
    net = inr.pack_mlp(params, inr.KIND_SIREN, 0, 4, w0=w0)
    c, f = torch.from_numpy(coords).cuda(), torch.from_numpy(feats).cuda()
    os.environ.pop("MRIRT_INR_NO_WS", None)
    got, cls = inr._forward(net, c, f, n, True, True)
    os.environ["MRIRT_INR_NO_WS"] = "1"
    ref, cls0 = inr._forward(net, c, f, n, True, True)
    os.environ.pop("MRIRT_INR_NO_WS", None)
    got, ref = got.cpu().numpy(), ref.cpu().numpy()
    scale = max(1.0, np.abs(want).max())
    print(n, "ws vs fp64", np.abs(got - want).max() / scale, "stream vs fp64", np.abs(ref - want).max() / scale,
          "ws vs stream", np.abs(got - ref).max() / scale, "argmax agree ws/stream", float((cls == cls0).float().mean()),
          "ws argmax vs fp64", float((cls.cpu().numpy() == want.argmax(1)).mean()), flush=True)
