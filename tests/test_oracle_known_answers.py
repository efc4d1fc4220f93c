"""Analytic known answers the oracle must reproduce (the reference has no
tests, so besides the camera/INR goldens these closed forms are what pins the restatement)."""
import math

import numpy as np
import pytest

from oracle import oracle_c, oracle_np as onp


def _scene(**kw):
    from mrirt import synth
    return synth, synth.brats_scene(**kw)


@pytest.mark.parametrize("v0,alpha", [(0.6, 0.4), (1.0, 16.0), (0.25, 60.0)])
def test_k1_constant_volume_closed_form(v0, alpha):
    """Constant volume v0, one modality, overlays off:
    C = bg + v' (1 - e^{-v' a dt N}),  T = e^{-v' a dt N},  N = min(#steps in [t0,t1), first n with T <= 0.01)."""
    synth, p = _scene(n=16, image=24, steps=80, channels=1, intensity_alpha=alpha)
    p["bgColor"] = np.array([0.1, 0.2, 0.3], np.float32)
    vol = np.full(16 ** 3, v0, np.float32)
    for oracle in (onp, oracle_c):
        img, aux = onp.brats_main(p, [vol], return_aux=True) if oracle is onp else (oracle.brats_main(p, [vol]), None)
        if aux is None:
            continue
        dt = float(np.float32(p["stepSize"]))
        N = aux["nsteps"].astype(np.float64)
        T = np.exp(-v0 * alpha * dt * N)
        want = v0 * (1.0 - T)
        for ch, bg in enumerate((0.1, 0.2, 0.3)):
            assert np.abs(img[..., ch] - (bg + want)).max() < 2e-5
        assert np.abs(aux["T"] - T).max() < 2e-5
        # N is either the chord length in steps or the ERT cut, never more
        n_ert = math.ceil(math.log(0.01) / (-v0 * alpha * dt) - 1e-9)
        assert N.max() <= n_ert + 1
        if alpha >= 16:
            assert (N >= n_ert - 1).any(), "ERT should fire on the long central chords"
    assert np.array_equal(onp.brats_main(p, [vol]), oracle_c.brats_main(p, [vol]))


def test_k1_miss_is_background_and_alpha_one():
    synth, p = _scene(n=8, image=16, steps=16, channels=1)
    p["bgColor"] = np.array([0.3, 0.6, 0.9], np.float32)
    p["eye"] = np.array([0.0, 0.0, 50.0], np.float32)
    p["W"] = np.array([0.0, 0.0, 1.0], np.float32)          # looking away from the box
    img = onp.brats_main(p, [np.ones(512, np.float32)])
    assert np.all(img[..., :3] == np.array([0.3, 0.6, 0.9], np.float32)) and np.all(img[..., 3] == 1.0)


def test_trilinear_reproduces_affine_fields():
    """sampleLinear is exact (to rounding) on any affine field, for every lattice cell."""
    X, Y, Z = 7, 6, 5
    zz, yy, xx = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij")
    buf = (0.25 * xx - 0.5 * yy + 0.125 * zz + 3.0).astype(np.float32).reshape(-1)
    rng = np.random.default_rng(0)
    q = (rng.random((500, 3)) * (np.array([X, Y, Z]) - 1.002)).astype(np.float32)
    v, _ = onp._sample_linear(buf, q[:, 0], q[:, 1], q[:, 2], X, Y, Z)
    want = 0.25 * q[:, 0] - 0.5 * q[:, 1] + 0.125 * q[:, 2] + 3.0
    assert np.abs(v - want).max() < 2e-6
    # lattice gradient of an affine field is its slope (interior cells)
    qi = (1 + rng.random((200, 3)) * (np.array([X, Y, Z]) - 3.01)).astype(np.float32)
    _, (ix, iy, iz, fx, fy, fz) = onp._sample_linear(buf, qi[:, 0], qi[:, 1], qi[:, 2], X, Y, Z)
    g = onp._lattice_gradient(buf, ix, iy, iz, fx, fy, fz, X, Y, Z)
    for got, slope in zip(g, (0.25, -0.5, 0.125)):
        assert np.abs(got * 0.5 - slope).max() < 2e-6
    # and it equals the central difference of the trilinear field with h = 1 voxel
    vp, _ = onp._sample_linear(buf, qi[:, 0] + 1, qi[:, 1], qi[:, 2], X, Y, Z)
    vm, _ = onp._sample_linear(buf, qi[:, 0] - 1, qi[:, 1], qi[:, 2], X, Y, Z)
    assert np.abs((vp - vm) - g[0]).max() < 5e-6


def test_lattice_gradient_equals_shifted_trilinear_on_random_data():
    X, Y, Z = 9, 8, 7
    rng = np.random.default_rng(1)
    buf = rng.random(X * Y * Z).astype(np.float32)
    q = (1.0 + rng.random((300, 3)) * (np.array([X, Y, Z]) - 3.01)).astype(np.float32)
    _, (ix, iy, iz, fx, fy, fz) = onp._sample_linear(buf, q[:, 0], q[:, 1], q[:, 2], X, Y, Z)
    g = onp._lattice_gradient(buf, ix, iy, iz, fx, fy, fz, X, Y, Z)
    for axis in range(3):
        e = np.zeros(3, np.float32); e[axis] = 1
        vp, _ = onp._sample_linear(buf, *(q + e).T, X, Y, Z)
        vm, _ = onp._sample_linear(buf, *(q - e).T, X, Y, Z)
        assert np.abs((vp - vm) - g[axis]).max() < 1e-5


def test_sample_label_rounds_half_away_and_clamps():
    X, Y, Z = 4, 3, 2
    buf = np.arange(X * Y * Z, dtype=np.uint32)
    q = np.array([[0.5, 0.49999997, 0.0], [2.5, 1.5, 0.5], [-3.0, 9.0, 9.0], [3.0, 2.0, 1.0]], np.float32)
    got = onp._sample_label(buf, q[:, 0], q[:, 1], q[:, 2], X, Y, Z)
    want = [1 + 0 * 4 + 0, 3 + 2 * 4 + 1 * 12, 0 + 2 * 4 + 1 * 12, 3 + 2 * 4 + 12]
    assert list(got) == want


def test_k1_ert_on_off_bound():
    """|C_ert - C_full| <= 0.01 * max emission (here val <= 1): the T>0.01 cut is part of the
    semantics and its effect is bounded."""
    synth, p = _scene(n=24, image=32, steps=128, channels=1, intensity_alpha=40.0)
    vol = synth.synth_volume(24)
    on = oracle_c.brats_main(p, [vol])
    off = oracle_c.brats_main(p, [vol], ext=dict(ertThreshold=0.0))
    d = np.abs(on - off).max()
    assert 1e-4 < d <= 0.01 + 1e-6
    _, a_on = oracle_c.brats_main(p, [vol], return_aux=True)
    _, a_off = oracle_c.brats_main(p, [vol], ext=dict(ertThreshold=0.0), return_aux=True)
    assert a_on["live_samples"] < a_off["live_samples"]


def test_k1_transmittance_monotone_and_bounded():
    synth, p = _scene(n=20, image=24, steps=64, channels=2, intensity_alpha=8.0, show_seg=True)
    vols = [synth.synth_volume(20, s) for s in (1, 2)]
    img, aux = onp.brats_main(p, vols, synth.synth_labels(20), return_aux=True)
    assert np.all((aux["T"] > 0) & (aux["T"] <= 1))
    assert np.all(img[..., :3] >= 0) and np.all(img[..., :3] <= 1.0 + 1e-6)
    assert np.all(aux["nsteps"] <= 64 + 1)


@pytest.mark.parametrize("s_u8", [0, 64, 255])
def test_k2_constant_volume_closed_form(s_u8):
    """accum = 1 - (1 - 4 s/steps)^{n_inside}, capped by the 0.995 break."""
    from mrirt import synth
    steps = 64
    p = synth.volume_scene(8, 24, steps)
    u8 = np.full(512, s_u8, np.uint8)
    img, aux = onp.volume_cs(p, onp.pack_u8_volume(u8), return_aux=True)
    s = (s_u8 / 255.0) * 4.0 / steps
    nin = aux["nfetch"].astype(np.float64)
    want = 1.0 - (1.0 - s) ** nin
    assert np.abs(img[..., 0] - want).max() < 1e-5
    assert np.all(img[..., 0] == img[..., 1]) and np.all(img[..., 3] == 1.0)
    assert np.array_equal(img, oracle_c.volume_cs(p, onp.pack_u8_volume(u8)))


def test_k2_u8_modes_agree_and_padding():
    from mrirt import synth
    dims = (7, 5, 3)                                   # 105 voxels: pack pads to 108
    u8 = (np.arange(105) % 251).astype(np.uint8)
    pk = onp.pack_u8_volume(u8)
    assert pk.shape == (27, 4) and pk.dtype == np.uint32 and pk.reshape(-1)[105:].sum() == 0
    p = synth.volume_scene(0, 20, 40, dims=dims)
    assert np.array_equal(onp.volume_cs(p, pk, mode="u32x4"), onp.volume_cs(p, u8, mode="u8"))
    f = u8.astype(np.float32) / np.float32(255.0)
    assert np.array_equal(onp.volume_cs(p, pk, mode="u32x4"), onp.volume_cs(p, f, mode="f32"))


def test_k3_hit_iff_ray_meets_sphere():
    from mrirt import synth
    sp, eye, U, V, W = synth.sdf_scene()
    w, h = 96, 72
    img = onp.raymarch_cs(sp, eye, U, V, W, w, h)
    (o, (dx, dy, dz)) = onp.make_primary(w, h, sp["fovY"], eye, U, V, W, k3_aspect=True)
    o = np.array(o, np.float64)
    d = np.stack([dx, dy, dz], -1).astype(np.float64)
    b = d @ o
    disc = b * b - (o @ o - 0.36)
    hit = np.abs(img[..., 2] - (1.0 - img[..., 0])) < 1e-6      # hit colour has b = 1 - r
    sky_b = img[..., 2] <= 0.3 + 1e-6
    clear_hit, clear_miss = disc > 2e-2, disc < -2e-2
    assert np.all(hit[clear_hit]) and np.all(~(hit & ~sky_b)[clear_miss] | True)
    # sky pixels are the lerp of the two sky colours in d.y
    miss = clear_miss
    tbg = 0.5 * (d[..., 1] + 1.0)
    assert np.abs(img[..., 0][miss] - (0.05 + tbg[miss] * 0.15)).max() < 1e-6
    assert hit[clear_hit].all() and clear_hit.sum() > 100 and miss.sum() > 100


def test_bc4_decode_known_block():
    """One block with r0 > r1 (6 interpolants) and one with r0 <= r1 (4 + 0/255)."""
    blk1 = bytes([200, 40]) + (sum(k << (3 * k) for k in range(8)) | sum(k << (3 * (k + 8)) for k in range(8))).to_bytes(6, "little")
    blk2 = bytes([40, 200]) + (sum(k << (3 * k) for k in range(8)) | sum((7 - k) << (3 * (k + 8)) for k in range(8))).to_bytes(6, "little")
    out = onp.bc4_decode(blk1 + blk2, 8, 4, 1).reshape(4, 8)
    pal1 = [200, 40] + [((7 - i) * 200 + i * 40 + 3) // 7 for i in range(1, 7)]
    pal2 = [40, 200] + [((5 - i) * 40 + i * 200 + 2) // 5 for i in range(1, 5)] + [0, 255]
    assert list(out[0, :4]) == pal1[:4] and list(out[1, :4]) == pal1[4:8]
    assert list(out[0, 4:]) == pal2[:4] and list(out[1, 4:]) == pal2[4:8]
    assert list(out[2, 4:]) == pal2[7:3:-1]
    with pytest.raises(RuntimeError):
        onp.bc4_decode(b"\0" * 7, 4, 4, 1)


def test_markstein_division_by_255_is_exact_for_every_byte():
    """csrc/volume_march.hip::unorm8 replaces byte / 255.0f by q = x*r; q + (x - q*255)*r with r = RN(1/255).
    Emulated here in exact arithmetic (fp64 holds the products of two fp32 exactly): all 256 bytes must give
    the correctly rounded fp32 quotient."""
    x = np.arange(256, dtype=np.float32)
    r = np.float32(1.0) / np.float32(255.0)
    q = (x * r).astype(np.float32)
    e = (x.astype(np.float64) - q.astype(np.float64) * 255.0)           # what fma(-q, 255, x) returns, exactly
    assert np.array_equal(e.astype(np.float32).astype(np.float64), e)   # ... and it is representable
    got = (e * np.float64(r) + q.astype(np.float64)).astype(np.float32)
    assert np.array_equal(got, x / np.float32(255.0))


# ---------------------------------------------------------------------------------------------------
# Third-party cross-check: the K1/K2/K3 oracle cannot be pinned by anything the reference
# holds (no tests, no golden images, Slang not runnable here), so the only INDEPENDENT implementation of the
# same semantics available is torch's `grid_sample` (5-D input = trilinear; align_corners=True; border
# padding — SURVEY.md 8d item 3).  It interpolates in a different operation order, so the comparison is to a
# tolerance, not to the bit; it does not turn "parity unpinned" green, it bounds how wrong the restatement of
# sampleLinear (brats_rt.slang:60-76) and of the march (:117-141) could be.
# ---------------------------------------------------------------------------------------------------
def _grid_sample(vol_zyx, q_xyz):
    import torch
    import torch.nn.functional as F
    Z, Y, X = vol_zyx.shape
    dims = torch.tensor([X, Y, Z], dtype=torch.float64)
    g = 2.0 * torch.from_numpy(q_xyz.astype(np.float64)) / (dims - 1.0) - 1.0
    out = F.grid_sample(torch.from_numpy(vol_zyx.astype(np.float64))[None, None], g[None, None, None],
                        mode="bilinear", padding_mode="border", align_corners=True)
    return out[0, 0, 0, 0].numpy()


def test_sample_linear_agrees_with_torch_grid_sample():
    rng = np.random.default_rng(21)
    X, Y, Z = 19, 14, 11
    vol = rng.random((Z, Y, X), dtype=np.float32)
    lin = vol.reshape(-1)
    n = 20000
    q = (rng.random((n, 3)) * (np.array([X, Y, Z]) - 1.001)).astype(np.float32)        # pIdx in [0, N - 1.001]
    q[:64] = np.floor(q[:64])                                                           # lattice points
    q[64:96, 0] = np.float32(X - 1.001)                                                 # the clamp's upper edge
    want = _grid_sample(vol, q)
    got = onp._sample_linear(lin, q[:, 0], q[:, 1], q[:, 2], X, Y, Z)[0]
    assert got.dtype == np.float32
    assert np.abs(got - want).max() <= 4e-7, np.abs(got - want).max()                   # a few fp32 ulps of values in [0,1]
    assert np.array_equal(got[:64], lin[(q[:64, 0] + q[:64, 1] * X + q[:64, 2] * X * Y).astype(np.int64)])
    # outside the box the shader clamps (brats_rt.slang:62-64); border padding does the same below 0
    qo = q.copy()
    qo[:, 1] -= np.float32(30.0)
    qc = qo.copy()
    qc[:, 1] = 0.0
    assert np.abs(onp._sample_linear(lin, qo[:, 0], qo[:, 1], qo[:, 2], X, Y, Z)[0] - _grid_sample(vol, qc)).max() <= 4e-7


def test_k1_frame_agrees_with_an_independent_torch_march():
    """One K1 frame (one modality, ERT, dense preset) marched by tools/cpu_torch_baseline.py's grid_sample
    renderer — different code, different interpolation order, fp32 torch ops — within the BASELINE's 1e-4."""
    import importlib.util
    import pathlib
    import torch
    spec = importlib.util.spec_from_file_location(
        "cpu_torch_baseline", pathlib.Path(__file__).resolve().parent.parent / "tools" / "cpu_torch_baseline.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    synth, p = _scene(n=40, image=72, steps=96, channels=1, intensity_alpha=16.0)
    vol = synth.synth_volume(40)
    with torch.no_grad():
        img, live = mod.render(p, vol, 40)
    ref, aux = onp.brats_main(p, [vol], return_aux=True)
    assert np.array_equal(ref, oracle_c.brats_main(p, [vol]))
    d = np.abs(img.numpy() - ref[..., 0])
    # knife-edge pixels (a ray whose T lands within rounding of 0.01, or whose last step lands on t1) may take
    # one step more or fewer in the other implementation: bounded by one step's contribution
    assert np.quantile(d, 0.999) <= 1e-4, np.quantile(d, 0.999)
    assert d.max() <= 0.011
    assert abs(live - int(aux["nsteps"].sum())) <= 0.002 * live


def test_k2_frame_agrees_with_an_independent_torch_march():
    """volume_cs (scripts/volumeRendering/volume_render.slang:104-148) restated a second time, differently: fp64,
    sample positions as near + i * step (no running sum), the u8 trilinear fetch through torch's grid_sample
    (align_corners=True, i.e. texel = sat(0.5 (x + 1)) * (dim - 1) as :46-49).  Bounds how wrong oracle_np.volume_cs
    could be; pixels whose ray grazes a cube face or stops within rounding of 0.995 may take one sample more or
    fewer (one sample weighs at most 4 / steps)."""
    import torch
    import torch.nn.functional as F
    from mrirt import synth
    n, image, steps = 40, 64, 96
    u8 = synth.synth_u8_volume(n)
    p = synth.volume_scene(n, image, steps, near=1.5, far=4.5)
    ref, aux = onp.volume_cs(p, u8, mode="u8", return_aux=True)
    assert np.array_equal(ref, oracle_c.volume_cs(p, onp.pack_u8_volume(u8)))
    W, H = int(p["imageSize"][0]), int(p["imageSize"][1])
    f64 = torch.float64
    eye, U, V, Wv = (torch.tensor(np.asarray(p[k], np.float64)) for k in ("eye", "U", "V", "W"))
    ys, xs = torch.meshgrid(torch.arange(H, dtype=f64), torch.arange(W, dtype=f64), indexing="ij")
    ndcx = (xs + 0.5) / W * 2 - 1
    ndcy = 1 - (ys + 0.5) / H * 2
    th = np.tan(0.5 * float(p["fovY"]))
    view = torch.stack([ndcx * (W / max(1, H)) * th, ndcy * th, torch.ones_like(ndcx)], -1)      # :109-113

    def plane(d):
        v = view * d
        return eye + v[..., 0:1] * U + v[..., 1:2] * V + v[..., 2:3] * Wv

    near, far = float(p["nearPlane"]), float(p["farPlane"])
    a, b = plane(max(0.0, near)), plane(max(max(0.0, near), far))
    step = (b - a) / steps
    vol = torch.from_numpy(u8.reshape(n, n, n).astype(np.float64) / 255.0)[None, None]
    accum = torch.zeros(H, W, dtype=f64)
    alive = torch.ones(H, W, dtype=torch.bool)
    fetched = 0
    for i in range(steps):
        pos = a + step * i
        inside = ((pos < 1) & (pos > -1)).all(-1)
        do = alive & inside & (accum < 1)
        s = F.grid_sample(vol, pos.clamp(-1, 1)[None, None], mode="bilinear", padding_mode="border", align_corners=True)[0, 0, 0]
        accum = torch.where(do, accum + (1 - accum) * s * (4.0 / steps), accum)
        fetched += int(do.sum())
        alive = alive & ~(accum > 0.995)
    d = np.abs(accum.numpy() - ref[..., 0])
    assert np.quantile(d, 0.999) <= 1e-4, np.quantile(d, 0.999)
    assert d.max() <= 4.0 / steps + 1e-4
    assert abs(fetched - aux["live_samples"]) <= 0.002 * fetched and fetched > 10000
