"""Generate the golden fixtures in this directory.  Runs ONLY in the build container, where the
reference checkout is mounted at /root/reference; the fixtures (data only) are committed, the
reference source never is.

  camera_yup.npz / camera_up.npz : produced by importing the reference's own
      scripts/raymarch/camera.py and inr/viewer/camera.py (pure NumPy) and recording
      (eye,U,V,W) for a table of states plus orbit/zoom/pan transitions.
  inr_fourier.npz                : produced by importing the reference's inr/inr/model.py
      and calling its build_input / apply_mlp / predict_volume on seeded fp32 parameters.
      model.py does `import jax, jax.numpy as jnp`; jax is not installed here, so the import
      is satisfied by binding the names `jax.numpy` -> numpy and `jax.nn.relu` -> np.maximum
      for the duration of the import (the four functions used are pure array expressions that
      read the same in either namespace; results are float64 where NumPy promotes, which the
      tests account for).  No reference code is copied or modified.
  inr_metrics.npz                : dice_score / coverage_dice of the same model.py (the names the reference's
      inr/interactive.ipynb imports next to model_load and predict_volume) on seeded label volumes,
      incl. a class that is absent from both volumes (NaN) and an all-background pair (0.0).
  siren.npz                      : produced by the reference's own `siren_apply`.  That function lives
      in a notebook code cell (notebooks/neumors_inr.ipynb, "SECTION 7" cell), not in an importable
      module: the generator reads the cell's JSON, parses it with `ast`, takes the one `FunctionDef`
      named siren_apply (decorator `@jit` included) and executes THAT node — nothing else of the
      cell — in a namespace where `jnp` is numpy and `jit` is the identity (the same binding used
      for model.py; the body is `jnp.sin`, `@` and `+` only).  Inputs are fed in fp64 (the notebook
      sets `jax_enable_x64`) and in fp32; both outputs are stored.  Networks: 3x256 and 4x256 with
      siren_init's zero biases, and the same with non-zero biases and a non-default w0.
      Data only is committed; the function text never is.
  render_*.npz                   : small oracle-rendered images (oracle_np) used as regression
      anchors for both oracles and for the HIP kernels — "oracle-defined", not reference-pinned.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py
"""
from __future__ import annotations

import importlib.util
import math
import os
import pathlib
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = pathlib.Path(__file__).resolve().parent
ROOT = HERE.parent.parent
REF = pathlib.Path(os.environ.get("MRIRT_REFERENCE", "/root/reference"))
sys.path.insert(0, str(ROOT))


def _load(path: pathlib.Path, name: str):
    spec = importlib.util.spec_from_file_location(name, str(path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def camera_goldens():
    ref_y = _load(REF / "scripts/raymarch/camera.py", "ref_camera_yup").OrbitalCamera
    ref_u = _load(REF / "inr/viewer/camera.py", "ref_camera_up").OrbitalCamera
    states = []
    rng = np.random.default_rng(7)
    fixed = [  # (target, radius, phi, theta)
        ((0, 0, 0), 3.0, math.radians(80), math.radians(25)),
        ((0, 0, 0), 4.2, math.radians(80), math.radians(25)),
        ((0.1, -0.2, 0.3), 2.0, math.pi * 0.5, 0.0),
        ((0, 0, 0), 2.0, 0.01, 1.0),                      # nearly looking straight down
        ((0, 0, 0), 2.0, math.pi - 0.01, -2.0),           # nearly straight up
        ((0, 0, 0), 2.0, 0.0, 0.0),                       # degenerate: forward || up
        ((0, 0, 0), 2.0, math.pi, 0.3),                   # degenerate the other way
        ((0, 0, 0), 0.0, 1.0, 1.0),                       # eye == target
        ((1.5, 2.5, -3.5), 7.25, 2.2, 4.4),
    ]
    for _ in range(7):
        fixed.append((tuple(rng.uniform(-1, 1, 3)), float(rng.uniform(0.2, 9)), float(rng.uniform(0.02, 3.1)),
                      float(rng.uniform(-7, 7))))
    ups = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1), (0.3, 0.9, -0.2)]
    out_y, out_u = [], []
    for tgt, rad, phi, th in fixed:
        c = ref_y(initial_target=np.array(tgt, dtype=np.float64), initial_radius=rad, initial_phi=phi, initial_theta=th)
        out_y.append(np.stack(c.get_basis()))
        for up in ups:
            c = ref_u(initial_target=np.array(tgt, dtype=np.float64), initial_radius=rad, initial_phi=phi,
                      initial_theta=th, world_up=np.array(up, dtype=np.float64))
            out_u.append(np.stack(c.get_basis()))
    st = np.array([list(t) + [r, p, h] for t, r, p, h in fixed], dtype=np.float64)

    # state-machine transitions: orbit / zoom / pan sequences
    ops = [("orbit", 0.3, -0.2), ("zoom", 1.7, 0), ("pan", 12.0, -7.0), ("orbit", -4.0, 5.0), ("zoom", 1e-3, 0),
           ("pan", -30.0, 44.0), ("zoom", 1e6, 0), ("orbit", 0.1, -9.0), ("pan", 3.0, 3.0)]
    trans_y, trans_u = [], []
    cy = ref_y(initial_radius=3.0, initial_phi=math.radians(80), initial_theta=math.radians(25))
    cu = ref_u(initial_radius=3.0, initial_phi=math.radians(80), initial_theta=math.radians(25),
               world_up=np.array([0.0, 0.0, 1.0]))
    for name, a, b in ops:
        for cam, rec, vh in ((cy, trans_y, None), (cu, trans_u, 480.0)):
            if name == "orbit":
                cam.orbit(a, b)
            elif name == "zoom":
                cam.zoom(a)
            elif vh is None:
                cam.pan(a, b)
            else:
                cam.pan(a, b, vh)
            rec.append(np.concatenate([np.stack(cam.get_basis()).reshape(-1),
                                       cam.target.astype(np.float64), [cam.radius, cam.phi, cam.theta]]))
    cy.reset(); cu.reset()
    trans_y.append(np.concatenate([np.stack(cy.get_basis()).reshape(-1), cy.target, [cy.radius, cy.phi, cy.theta]]))
    trans_u.append(np.concatenate([np.stack(cu.get_basis()).reshape(-1), cu.target, [cu.radius, cu.phi, cu.theta]]))
    opcodes = np.array([[{"orbit": 0, "zoom": 1, "pan": 2}[n], a, b] for n, a, b in ops], dtype=np.float64)
    np.savez(HERE / "camera_yup.npz", states=st, basis=np.stack(out_y).astype(np.float32),
             ops=opcodes, trans=np.stack(trans_y))
    np.savez(HERE / "camera_up.npz", states=st, ups=np.array(ups, dtype=np.float64),
             basis=np.stack(out_u).astype(np.float32).reshape(len(fixed), len(ups), 4, 3),
             ops=opcodes, trans=np.stack(trans_u), pan_viewport=480.0)
    print("camera goldens:", len(fixed), "states x", len(ups), "ups")


def _import_reference_model():
    """Import inr/inr/model.py with `jax.numpy` bound to numpy (see module docstring)."""
    jax = types.ModuleType("jax")
    jnp = types.ModuleType("jax.numpy")
    jnp.__dict__.update({k: getattr(np, k) for k in dir(np) if not k.startswith("_")})
    jnp.ndarray = np.ndarray
    nn = types.ModuleType("jax.nn")
    nn.relu = lambda x: np.maximum(x, 0)
    jax.numpy, jax.nn = jnp, nn
    jax.jit = lambda f, **k: f
    saved = {k: sys.modules.get(k) for k in ("jax", "jax.numpy", "jax.nn")}
    sys.modules.update({"jax": jax, "jax.numpy": jnp, "jax.nn": nn})
    try:
        return _load(REF / "inr/inr/model.py", "ref_inr_model")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def metrics_goldens():
    model = _import_reference_model()
    rng = np.random.default_rng(77)
    out = {}
    cases = []
    a = rng.integers(0, 4, (9, 8, 7)).astype(np.int16); b = rng.integers(0, 4, (9, 8, 7)).astype(np.int16)
    cases.append((a, b, 4))
    c = np.where(rng.random((6, 6, 6)) < 0.7, 0, rng.integers(1, 3, (6, 6, 6))).astype(np.int16)      # class 3 absent: NaN
    d = np.where(rng.random((6, 6, 6)) < 0.6, 0, rng.integers(1, 3, (6, 6, 6))).astype(np.int16)
    cases.append((c, d, 4))
    cases.append((np.zeros((4, 4, 4), np.int16), np.zeros((4, 4, 4), np.int16), 3))                     # nothing but background
    cases.append((a, a.copy(), 5))                                                                        # identical; class 4 absent
    for i, (pred, true, nc) in enumerate(cases):
        sc = model.dice_score(pred, true, nc)
        out[f"c{i}_pred"], out[f"c{i}_true"], out[f"c{i}_nc"] = pred, true, np.int64(nc)
        out[f"c{i}_dice"] = np.array([float(sc[k]) for k in range(nc)], dtype=np.float64)
        out[f"c{i}_coverage"] = np.float64(model.coverage_dice(pred, true))
    out["n"] = np.int64(len(cases))
    np.savez_compressed(HERE / "inr_metrics.npz", **out)
    print("metrics goldens:", len(cases), "cases")


def inr_goldens():
    model = _import_reference_model()
    rng = np.random.default_rng(2024)
    out = {}
    H, W, D, M = 6, 5, 4, 4
    mods = rng.standard_normal((M, H, W, D)).astype(np.float32)
    out["mods"] = mods
    for tag, K, hidden in (("k4h64", 4, [64] * 4), ("k16h256", 16, [256] * 4), ("k2h32x2", 2, [32, 32])):
        in_dim = 3 + 6 * K + M
        dims = [in_dim] + hidden + [4]
        params = []
        for i in range(len(dims) - 1):
            lim = math.sqrt(6.0 / (dims[i] + dims[i + 1]))
            params.append({"W": rng.uniform(-lim, lim, (dims[i], dims[i + 1])).astype(np.float32),
                           "b": rng.uniform(-0.1, 0.1, dims[i + 1]).astype(np.float32)})
        coords = rng.uniform(-1, 1, (37, 3)).astype(np.float32)
        feats = rng.standard_normal((37, M)).astype(np.float32)
        x = model.build_input(coords, feats, K)
        logits = model.apply_mlp(params, x)
        pred, _ = model.predict_volume(params, {"mods": mods, "seg": None}, K, chunk=50)
        for i, p in enumerate(params):
            out[f"{tag}_W{i}"], out[f"{tag}_b{i}"] = p["W"], p["b"]
        out[f"{tag}_K"] = np.int64(K)
        out[f"{tag}_coords"], out[f"{tag}_feats"] = coords, feats
        out[f"{tag}_x"] = np.asarray(x)
        out[f"{tag}_logits"] = np.asarray(logits)
        out[f"{tag}_pred"] = np.asarray(pred)
        print("inr golden", tag, "x", np.asarray(x).shape, np.asarray(x).dtype, "pred", pred.shape, pred.dtype)
    np.savez_compressed(HERE / "inr_fourier.npz", **out)


def _reference_siren_apply():
    """The notebook's own `siren_apply` (neumors_inr.ipynb:1165-1178), executed from the notebook file:
    cell JSON -> ast -> that single FunctionDef, with `jnp` = numpy and `jit` = identity."""
    import ast
    import json
    nb = json.loads((REF / "notebooks/neumors_inr.ipynb").read_text())
    for cell in nb["cells"]:
        src = "".join(cell.get("source", []))
        if cell.get("cell_type") != "code" or "def siren_apply" not in src:
            continue
        tree = ast.parse(src)
        fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "siren_apply"]
        if len(fn) != 1:
            continue
        ns = {"jnp": np, "jit": lambda f, **k: f}
        exec(compile(ast.Module(body=fn, type_ignores=[]), "<neumors_inr.ipynb:siren_apply>", "exec"), ns)
        return ns["siren_apply"]
    raise SystemExit("siren_apply not found in notebooks/neumors_inr.ipynb")


def siren_goldens():
    ref_apply = _reference_siren_apply()
    rng = np.random.default_rng(99)
    out = {"generator": np.array("reference: notebooks/neumors_inr.ipynb siren_apply (ast-extracted, jnp=numpy)")}
    cases = (("s3x256", 3, 30.0, False), ("s4x256", 4, 30.0, False),
             ("s3x256b", 3, 30.0, True), ("s4x256b", 4, 30.0, True), ("s4x256w", 4, 12.5, True))
    for tag, depth, w0, biased in cases:
        dims = [7] + [256] * depth + [4]
        params = {}
        for i in range(len(dims) - 1):
            lim = math.sqrt(6.0 / dims[i]) / (w0 if i == 0 else 1.0)     # siren_init, neumors_inr.ipynb:1150-1163
            b = rng.uniform(-0.3, 0.3, dims[i + 1]).astype(np.float32) if biased else np.zeros(dims[i + 1], np.float32)
            params[f"l{i}"] = {"w": rng.uniform(-lim, lim, (dims[i], dims[i + 1])).astype(np.float32), "b": b}
        x = np.concatenate([rng.uniform(-1, 1, (41, 3)), rng.uniform(0, 1, (41, 4))], axis=1).astype(np.float32)
        p64 = {k: {"w": v["w"].astype(np.float64), "b": v["b"].astype(np.float64)} for k, v in params.items()}
        logits = np.asarray(ref_apply(p64, x.astype(np.float64), w0=w0))        # x64, as the notebook runs it
        logits32 = np.asarray(ref_apply(params, x, w0=np.float32(w0)))          # the same function on fp32 arrays
        assert logits.dtype == np.float64 and logits.shape == (41, 4)
        for k, v in params.items():
            out[f"{tag}_{k}_w"], out[f"{tag}_{k}_b"] = v["w"], v["b"]
        out[f"{tag}_x"], out[f"{tag}_logits"], out[f"{tag}_logits32"] = x, logits, logits32.astype(np.float32)
        out[f"{tag}_w0"] = np.float64(w0)
        print("siren golden", tag, "logit range", float(np.abs(logits).max()), "fp32-vs-x64", float(np.abs(logits32 - logits).max()))
    np.savez_compressed(HERE / "siren.npz", **out)


def render_goldens():
    from oracle import oracle_np as onp
    import mrirt
    from mrirt import synth
    n = 20
    vols = [synth.synth_volume(n, 1234 + m, phase=0.3 * m, dims=(20, 18, 14)) for m in range(4)]
    lab = synth.synth_labels(n, dims=(20, 18, 14))
    out = {}
    cases = {
        "k1_default": (dict(channels=4, show_seg=True), None),
        "k1_dense_pred": (dict(channels=2, show_seg=True, show_pred=True, intensity_alpha=16.0), None),
        "k1_shade": (dict(channels=1, intensity_alpha=16.0), synth.SHADE_EXT),
    }
    for name, (kw, ext) in cases.items():
        p = synth.brats_scene(n, 40, 64, dims=(20, 18, 14), image_hw=(36, 40), **kw)
        img, aux = onp.brats_main(p, vols, lab, lab[::-1].copy(), ext, return_aux=True)
        out[name] = img
        out[name + "_live"] = np.int64(aux["live_samples"])
    u8 = np.rint(vols[0] * 255).astype(np.uint8)
    p = synth.volume_scene(n, 40, 48, dims=(20, 18, 14))
    out["k2_u8"] = onp.volume_cs(p, onp.pack_u8_volume(u8), mode="u32x4")
    sp, eye, U, V, W = synth.sdf_scene()
    out["k3"] = onp.raymarch_cs(sp, eye, U, V, W, 40, 36)
    np.savez_compressed(HERE / "render_small.npz", **out)
    print("render goldens:", sorted(out))


if __name__ == "__main__":
    if not REF.exists():
        raise SystemExit(f"{REF} not found: goldens are generated in the build container only")
    if len(sys.argv) > 1 and sys.argv[1] == "metrics":       # (only the newest fixture: the others stay byte-identical)
        metrics_goldens()
        raise SystemExit(0)
    camera_goldens()
    inr_goldens()
    metrics_goldens()
    siren_goldens()
    render_goldens()
