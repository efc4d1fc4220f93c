"""Pin the oracle (and the product's host-side mirrors) to fixtures captured from the
reference's own importable Python — tests/golden/make_goldens.py says how each was made."""
import math

import numpy as np
import pytest

from oracle import oracle_c, oracle_np as onp


def _cams():
    import mrirt
    return mrirt.camera.OrbitalCamera


def _apply(cam, op, a, b, viewport):
    if op == 0:
        cam.orbit(a, b)
    elif op == 1:
        cam.zoom(a)
    elif viewport is None:
        cam.pan(a, b)
    else:
        cam.pan(a, b, viewport)


def _state_vec(cam):
    return np.concatenate([np.stack(cam.get_basis()).reshape(-1), np.asarray(cam.target, np.float64),
                           [cam.radius, cam.phi, cam.theta]])


def test_camera_yup_matches_reference(golden_dir):
    g = np.load(golden_dir / "camera_yup.npz")
    Prod = _cams()
    for st, basis in zip(g["states"], g["basis"]):
        tgt, rad, phi, th = st[:3], st[3], st[4], st[5]
        o = onp.OrbitalCameraY(target=tgt, radius=rad, phi=phi, theta=th)
        p = Prod(initial_target=np.array(tgt), initial_radius=rad, initial_phi=phi, initial_theta=th)
        for cam in (o, p):
            got = np.stack(cam.get_basis())
            assert got.dtype == np.float32
            assert np.array_equal(got, basis, equal_nan=True), (st, type(cam).__name__)


def test_camera_worldup_matches_reference(golden_dir):
    g = np.load(golden_dir / "camera_up.npz")
    Prod = _cams()
    for st, per_up in zip(g["states"], g["basis"]):
        tgt, rad, phi, th = st[:3], st[3], st[4], st[5]
        for up, basis in zip(g["ups"], per_up):
            o = onp.OrbitalCameraUp(target=tgt, radius=rad, phi=phi, theta=th, world_up=up)
            p = Prod(initial_target=np.array(tgt), initial_radius=rad, initial_phi=phi, initial_theta=th,
                     world_up=np.array(up))
            for cam in (o, p):
                assert np.array_equal(np.stack(cam.get_basis()), basis, equal_nan=True), (st, up, type(cam).__name__)


@pytest.mark.parametrize("which", ["camera_yup.npz", "camera_up.npz"])
def test_camera_state_machine_matches_reference(golden_dir, which):
    g = np.load(golden_dir / which)
    Prod = _cams()
    kw = dict(initial_radius=3.0, initial_phi=math.radians(80), initial_theta=math.radians(25))
    if which == "camera_up.npz":
        viewport = float(g["pan_viewport"])
        cams = [onp.OrbitalCameraUp(radius=3.0, phi=math.radians(80), theta=math.radians(25), world_up=[0, 0, 1]),
                Prod(world_up=np.array([0.0, 0.0, 1.0]), **kw)]
    else:
        viewport = None
        cams = [onp.OrbitalCameraY(radius=3.0, phi=math.radians(80), theta=math.radians(25)), Prod(**kw)]
    for cam in cams:
        for (op, a, b), want in zip(g["ops"], g["trans"]):
            _apply(cam, int(op), a, b, viewport)
            assert np.array_equal(_state_vec(cam), want), (type(cam).__name__, op, a, b)
    # reset() restores the initial frame (last golden row); only the product mirrors reset()
    cams[1].reset()
    assert np.array_equal(_state_vec(cams[1]), g["trans"][-1])


@pytest.mark.parametrize("tag,nlayers", [("k4h64", 5), ("k16h256", 5), ("k2h32x2", 3)])
def test_inr_fourier_matches_reference(golden_dir, tag, nlayers):
    g = np.load(golden_dir / "inr_fourier.npz")
    K = int(g[f"{tag}_K"])
    params = [{"W": g[f"{tag}_W{i}"], "b": g[f"{tag}_b{i}"]} for i in range(nlayers)]
    x = onp.build_input(g[f"{tag}_coords"], g[f"{tag}_feats"], K)
    # the reference (run with jnp -> numpy) promotes the integer frequencies to float64; the
    # oracle is fp32 like JAX: feature ORDER must match exactly, values to fp32 rounding
    assert x.shape == g[f"{tag}_x"].shape and x.dtype == np.float32
    # |arg| <= K*pi (50 at K=16): one fp32 ulp of the argument is 4e-6
    assert np.abs(x - g[f"{tag}_x"]).max() < 2e-6 * max(4, K)
    logits = onp.apply_mlp(params, x)
    assert np.abs(logits - g[f"{tag}_logits"]).max() < 5e-5 * max(1.0, np.abs(g[f"{tag}_logits"]).max())
    pred, _ = onp.predict_volume(params, {"mods": g["mods"], "seg": None}, K, chunk=50)
    assert pred.dtype == np.int16 and pred.shape == g[f"{tag}_pred"].shape
    assert np.array_equal(pred, g[f"{tag}_pred"])


SIREN_CASES = (("s3x256", 3), ("s4x256", 4), ("s3x256b", 3), ("s4x256b", 4), ("s4x256w", 4))


def test_siren_oracle_fixture(golden_dir):
    """reference-pinned: the fixture holds the output of the notebook's own siren_apply
    (neumors_inr.ipynb:1165-1178, extracted with ast by make_goldens.py) on fp64 and on fp32 arrays, zero and
    non-zero biases, default and non-default w0.  The fp32 restatement must reproduce both."""
    g = np.load(golden_dir / "siren.npz")
    assert str(g["generator"]).startswith("reference:")
    for tag, depth in SIREN_CASES:
        params = {f"l{i}": {"w": g[f"{tag}_l{i}_w"], "b": g[f"{tag}_l{i}_b"]} for i in range(depth + 1)}
        out = onp.siren_apply(params, g[f"{tag}_x"], w0=float(g[f"{tag}_w0"]))
        assert out.shape == (41, 4)
        assert np.abs(out - g[f"{tag}_logits"]).max() < 2e-5          # x64 run of the reference; fp32 noise is ~1e-6
        assert np.abs(out - g[f"{tag}_logits32"]).max() < 2e-5
        assert np.array_equal(out.argmax(-1), g[f"{tag}_logits"].argmax(-1))


def _small_scene():
    from mrirt import synth
    dims = (20, 18, 14)
    vols = [synth.synth_volume(0, 1234 + m, phase=0.3 * m, dims=dims) for m in range(4)]
    lab = synth.synth_labels(0, dims=dims)
    return synth, dims, vols, lab


def test_render_regression_fixtures(golden_dir):
    """oracle-defined regression anchors: both restatements reproduce the committed images
    bit for bit (guards the oracle itself against silent edits)."""
    g = np.load(golden_dir / "render_small.npz")
    synth, dims, vols, lab = _small_scene()
    cases = {
        "k1_default": (dict(channels=4, show_seg=True), None),
        "k1_dense_pred": (dict(channels=2, show_seg=True, show_pred=True, intensity_alpha=16.0), None),
        "k1_shade": (dict(channels=1, intensity_alpha=16.0), synth.SHADE_EXT),
    }
    for name, (kw, ext) in cases.items():
        p = synth.brats_scene(20, 40, 64, dims=dims, image_hw=(36, 40), **kw)
        a, aux = onp.brats_main(p, vols, lab, lab[::-1].copy(), ext, return_aux=True)
        b, auxc = oracle_c.brats_main(p, vols, lab, lab[::-1].copy(), ext, return_aux=True)
        assert np.array_equal(a, g[name]) and np.array_equal(b, g[name]), name
        assert aux["live_samples"] == auxc["live_samples"] == int(g[name + "_live"])
    u8 = np.rint(vols[0] * 255).astype(np.uint8)
    p = synth.volume_scene(20, 40, 48, dims=dims)
    assert np.array_equal(onp.volume_cs(p, onp.pack_u8_volume(u8), mode="u32x4"), g["k2_u8"])
    assert np.array_equal(oracle_c.volume_cs(p, onp.pack_u8_volume(u8), mode="u32x4"), g["k2_u8"])
    sp, eye, U, V, W = synth.sdf_scene()
    assert np.array_equal(onp.raymarch_cs(sp, eye, U, V, W, 40, 36), g["k3"])
    assert np.array_equal(oracle_c.raymarch_cs(sp, eye, U, V, W, 40, 36), g["k3"])


@pytest.mark.parametrize("seed", range(4))
def test_two_restatements_agree_bitwise(seed):
    """NumPy and C oracles are independent restatements of the same shader text: random
    cameras, dims, flags — images and live-sample counts must be identical."""
    import mrirt
    from mrirt import synth
    rng = np.random.default_rng(seed)
    dims = tuple(int(v) for v in rng.integers(5, 28, 3))
    vols = [synth.synth_volume(0, seed * 10 + m, phase=float(rng.uniform(0, 3)), dims=dims) for m in range(4)]
    lab = synth.synth_labels(0, dims=dims)
    cam = synth.bench_camera(radius=float(rng.uniform(1.2, 4)), phi_deg=float(rng.uniform(5, 175)),
                             theta_deg=float(rng.uniform(0, 360)))
    p = synth.brats_scene(0, 0, int(rng.integers(8, 80)), dims=dims, image_hw=(int(rng.integers(3, 40)), int(rng.integers(3, 40))),
                          channels=int(rng.integers(0, 5)), show_seg=bool(rng.integers(0, 2)), show_pred=bool(rng.integers(0, 2)),
                          intensity_alpha=float(rng.choice([0.4, 4.0, 16.0, 60.0])), camera=cam, fov_deg=float(rng.uniform(20, 90)))
    p["nearT"], p["farT"] = float(rng.choice([0.0, 1.0])), float(rng.choice([0.0, 3.0]))
    p["gamma"] = float(rng.choice([1.0, 0.7, 2.2]))
    p["bgColor"] = rng.random(3).astype(np.float32)
    p["volWeight"] = tuple(float(v) for v in rng.uniform(0.2, 2.0, 4))
    ext = dict(synth.SHADE_EXT, cameraMode=int(rng.integers(0, 2))) if rng.integers(0, 2) else None
    a, aux = onp.brats_main(p, vols, lab, np.roll(lab, 3), ext, return_aux=True)
    b, auxc = oracle_c.brats_main(p, vols, lab, np.roll(lab, 3), ext, return_aux=True)
    assert np.array_equal(a, b, equal_nan=True)
    assert aux["live_samples"] == auxc["live_samples"] and aux["shaded_samples"] == auxc["shaded_samples"]
    # row-band rendering is the same image
    H = int(p["imageSize"][1])
    if H >= 4:
        band = oracle_c.brats_main(p, vols, lab, np.roll(lab, 3), ext, rows=(1, H - 1))
        assert np.array_equal(band, b[1:H - 1], equal_nan=True)


def test_dice_scores_match_the_reference(golden_dir):
    """inr.dice_score / inr.coverage_dice (the names inr/interactive.ipynb imports from inr.model) against values produced by
    the reference's own model.py (tests/golden/make_goldens.py::metrics_goldens): NumPy inputs and torch tensors, a class
    absent from both volumes (NaN), an all-background pair (coverage 0.0)."""
    import torch
    from mrirt import inr
    g = np.load(golden_dir / "inr_metrics.npz")
    for i in range(int(g["n"])):
        pred, true, nc = g[f"c{i}_pred"], g[f"c{i}_true"], int(g[f"c{i}_nc"])
        want = g[f"c{i}_dice"]
        for p, t in ((pred, true), (torch.from_numpy(pred), torch.from_numpy(true)), (torch.from_numpy(pred), true)):
            got = inr.dice_score(p, t, nc)
            assert sorted(got) == list(range(nc))
            np.testing.assert_array_equal(np.array([got[c] for c in range(nc)]), want)      # same expression, same bits (NaN == NaN here)
            assert inr.coverage_dice(p, t) == float(g[f"c{i}_coverage"])
    with pytest.raises(ValueError):
        inr.dice_score(np.zeros(5, np.int16), np.zeros(6, np.int16), 2)
