"""The headless viewer end to end on a synthetic BraTS-shaped case written as .nii.gz files:
ingest -> normalise -> upload -> frame_volume -> dispatch, against the oracle fed the same
arrays; plus the INR prepass overlay (on_click_load_inr flow)."""
import json
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def case(tmp_path_factory):
    from mrirt import nifti
    d = tmp_path_factory.mktemp("BraTS-GLI-00000-000")
    rng = np.random.default_rng(8)
    dims = (30, 26, 20)
    x, y, z = np.meshgrid(*[np.linspace(-1, 1, n) for n in dims], indexing="ij")
    r = np.sqrt(x * x + y * y + z * z)
    raw = {}
    for i, suf in enumerate(("t1n", "t1c", "t2w", "t2f")):
        v = np.clip(1.2 - r, 0, None) * (800 + 100 * i) * (1 + 0.2 * np.sin((4 + i) * x)) + rng.random(dims) * 40
        v[r > 1.1] = 0.0
        raw[suf] = v.astype(np.float32)
        nifti.write_nifti(d / f"BraTS-GLI-00000-000-{suf}.nii.gz", raw[suf].astype(np.int16), zooms=(1.0, 1.0, 1.0))
        raw[suf] = raw[suf].astype(np.int16).astype(np.float32)
    seg = np.zeros(dims, np.uint8)
    seg[r < 0.5], seg[r < 0.35], seg[r < 0.2] = 2, 1, 3
    nifti.write_nifti(d / "BraTS-GLI-00000-000-seg.nii.gz", seg)
    return d, dims, raw, seg


def test_viewer_frame_matches_oracle(case):
    import mrirt
    from mrirt.viewer import BraTSViewer
    from oracle import oracle_c, oracle_np as onp
    d, dims, raw, seg = case
    v = BraTSViewer(d, up="Z")
    assert tuple(v.vol_dims) == dims and v.seg_buffer is not None
    assert abs(v.camera.radius - float(np.linalg.norm(v.voxel_size * np.array(dims, np.float32)) * 0.8)) < 1e-6
    v.camera.orbit(0.4, -0.3)
    v.step_size = 0.02
    tex = v.render(96, 64)
    got = tex.to_numpy()
    assert got.dtype == np.float16 and got.shape == (64, 96, 4)
    # the oracle, fed by the oracle's own prep of the same arrays and the oracle's camera
    vols = [onp.normalize_volume(raw[s])[0] for s in ("t1n", "t1c", "t2w", "t2f")]
    lab = onp.flatten_labels(seg.astype(np.float32))[0]
    p = v.params(96, 64)
    cam = onp.OrbitalCameraUp(target=v.camera.target, radius=v.camera.radius, phi=v.camera.phi, theta=v.camera.theta,
                              world_up=[0, 0, 1])
    for a, b in zip(cam.get_basis(), (p["eye"], p["U"], p["V"], p["W"])):
        assert np.array_equal(a, b)
    ref = oracle_c.brats_main(p, vols, lab, None)
    assert np.array_equal(got, ref.astype(np.float16))
    assert ref[..., :3].max() > 0.05, "the case must actually be visible"
    # frame loop with auto-rotate + PNG dump
    out = d / "frames"
    out.mkdir()
    last = v.run(3, 48, 32, d_theta=math.radians(5), out_dir=out)
    assert last.shape == (32, 48, 4) and len(list(out.glob("frame_*.png"))) == 3


def test_viewer_inr_prepass_overlay(case, tmp_path):
    import mrirt
    from mrirt.viewer import BraTSViewer, MOD_ORDER
    from oracle import oracle_c, oracle_np as onp
    d, dims, raw, seg = case
    rng = np.random.default_rng(3)
    K = 4
    sizes = [3 + 6 * K + 4] + [64] * 4 + [4]
    flat = {}
    for i in range(5):
        flat[f"W_{i}"] = (rng.uniform(-1, 1, (sizes[i], sizes[i + 1])) * np.sqrt(6 / (sizes[i] + sizes[i + 1]))).astype(np.float32)
        flat[f"b_{i}"] = rng.uniform(-0.3, 0.3, sizes[i + 1]).astype(np.float32)
    np.savez(tmp_path / "inr.npz", **flat)
    (tmp_path / "inr_info.json").write_text(json.dumps({"config": {"FOURIER_FREQS": K}}))
    v = BraTSViewer(d, up="Y")
    v.load_inr(tmp_path / "inr.npz")
    assert v.show_pred and v.pred_buffer is not None
    pred_lin = v.pred_buffer.tensor.cpu().numpy()
    # the prepass labels agree with the fp32 oracle's predict_volume on the same z-scored inputs
    mods = np.stack([onp.zscore_modality(v.raw_volumes[m]) for m in MOD_ORDER], 0)
    params = [{"W": flat[f"W_{i}"], "b": flat[f"b_{i}"]} for i in range(5)]
    want, _ = onp.predict_volume(params, {"mods": mods, "seg": None}, K)
    want_lin = want.transpose(2, 1, 0).reshape(-1)
    assert (pred_lin == want_lin).mean() >= 0.99
    # and the frame is exactly the oracle's frame for the label grid the GPU produced
    v.step_size = 0.03
    got = v.render(64, 48).to_numpy()
    vols = [onp.normalize_volume(raw[s])[0] for s in ("t1n", "t1c", "t2w", "t2f")]
    lab = onp.flatten_labels(seg.astype(np.float32))[0]
    ref = oracle_c.brats_main(v.params(64, 48), vols, lab, pred_lin.astype(np.uint32))
    assert np.array_equal(got, ref.astype(np.float16))
