"""Edges real volumes can reach: grids beyond 32-bit byte offsets — the `wide` VG fall-back, label grids
and LINEAR grids of 2^30 elements — and voxels that are not finite or not normal, through the STRICT arithmetic, against the
oracle bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def slicewise_volume(dims, seed):
    """fp32 texture in [0, 1] generated slice by slice (a 2^30-voxel grid must not cost 3 x 8 GiB of meshgrid)."""
    X, Y, Z = dims
    rng = np.random.default_rng(seed)
    xs = np.linspace(-1, 1, X, dtype=np.float32)[None, :]
    ys = np.linspace(-1, 1, Y, dtype=np.float32)[:, None]
    out = np.empty((Z, Y, X), np.float32)
    for k, z in enumerate(np.linspace(-1, 1, Z, dtype=np.float32)):
        r = np.sqrt(xs * xs + ys * ys + z * z)
        out[k] = np.clip((1.0 - r) * (0.75 + 0.25 * np.sin(9 * xs) * np.cos(7 * ys)), 0, 1) + np.float32(0.04) * rng.random((Y, X), dtype=np.float32)
    return out.reshape(-1)


def test_wide_vg_grid_takes_the_generic_kernel_and_matches_the_oracle():
    """1024 x 1024 x 272 float4 voxels = 4.6 GB: byte offsets need 64 bits (GridDims::wide), the pipelined kernels step aside."""
    import torch
    import mrirt
    from mrirt import render, synth
    from oracle import oracle_c
    dims = (1024, 1024, 272)
    vol = slicewise_volume(dims, 3)
    p = synth.brats_scene(0, 0, 320, dims=dims, image_hw=(160, 224), channels=1, intensity_alpha=9.0)
    ext = dict(synth.SHADE_EXT, layout="vg")
    assert render.kernel_family(p, ext)["family"] == "generic"
    g = mrirt.upload_grid(vol, dims, "vg")
    assert g.nbytes >= 1 << 32
    got, st = mrirt.render_brats(p, [g], ext=ext, stats=True)
    ref, aux = oracle_c.brats_main(p, [vol], None, None, synth.SHADE_EXT, return_aux=True)
    assert np.array_equal(got.cpu().numpy(), ref) and st["live_samples"] == aux["live_samples"] > 0
    # skip=True on such a grid is the plain launch (no skipping kernel beyond 32-bit offsets): same bits
    assert torch.equal(mrirt.render_brats(p, [g], ext=ext, skip=True), got)
    # the far corner of the grid is really addressed: a frame that looks at it from outside the +x +y +z corner
    cam = mrirt.OrbitalCamera(initial_radius=2.2, initial_phi=np.radians(50), initial_theta=np.radians(40))
    eye, U, V, W = cam.get_basis()
    q = dict(p, eye=eye, U=U, V=V, W=W)
    assert np.array_equal(mrirt.render_brats(q, [g], ext=ext).cpu().numpy(), oracle_c.brats_main(q, [vol], None, None, synth.SHADE_EXT))
    del g
    # the QUAD form of the same volume is 4.6 GB too
    gq = mrirt.upload_grid(vol, dims, "quad")
    assert render.kernel_family(p, dict(layout="quad"))["family"] == "generic"
    assert np.array_equal(mrirt.render_brats(p, [gq], ext=dict(layout="quad")).cpu().numpy(), oracle_c.brats_main(p, [vol], None, None, None))


def test_label_and_linear_grids_of_2_to_the_30_elements():
    """1024^3: the label grid has 2^30 words and the LINEAR intensity grid 2^30 voxels — both past what the pipelined kernels'
    32-bit byte offsets reach (launch()): generic kernel, oracle bits, incl. the last voxel of both grids."""
    import mrirt
    from mrirt import render, synth
    from oracle import oracle_c
    dims = (1024, 1024, 1024)
    vol = slicewise_volume(dims, 5)
    lab = np.zeros((1024, 1024, 1024), np.uint32)
    lab[380:640, 300:700, 350:720] = 1
    lab[470:560, 460:580, 480:560] = 3
    lab[1000:, 990:, 985:] = 2                     # a block that includes the last voxel (index 2^30 - 1)
    lab = lab.reshape(-1)
    p = synth.brats_scene(0, 0, 300, dims=dims, image_hw=(120, 144), channels=1, show_seg=True, intensity_alpha=2.0)
    assert render.kernel_family(p, dict(layout="linear"))["family"] == "generic"
    g = mrirt.upload_grid(vol, dims, "linear", macro=False)
    gl = mrirt.upload_grid(lab, dims, "linear", macro=False)
    for cam in (None, mrirt.OrbitalCamera(initial_radius=2.0, initial_phi=np.radians(52), initial_theta=np.radians(38))):
        q = dict(p)
        if cam is not None:
            q["eye"], q["U"], q["V"], q["W"] = cam.get_basis()
        got, st = mrirt.render_brats(q, [g], labels=gl, stats=True)
        ref, aux = oracle_c.brats_main(q, [vol], lab, None, None, return_aux=True)
        assert np.array_equal(got.cpu().numpy(), ref) and st["live_samples"] == aux["live_samples"] > 0


@pytest.mark.parametrize("layout,shade", [("linear", False), ("brick", False), ("quad", False), ("vg", False), ("vg", True), ("vga", True), ("brick", True)])
def test_non_finite_and_denormal_voxels_through_strict_math(layout, shade):
    """+inf, -inf, NaN and denormal voxels sprinkled through a volume (the reference's loader clips to [0, 1], so this is an
    edge — but the claimed STRICT bit-identity has to hold there too): the window test `saturate((v - lo) / ww)`
    of an infinite sample is 1 with a true division and was 0 through Markstein's residual (inf - inf).  Whole frame and
    live-sample count against the oracle, plain and skipping launches."""
    import torch
    import mrirt
    from mrirt import synth
    from oracle import oracle_c
    dims = (44, 40, 36)
    rng = np.random.default_rng(17)
    vol = synth.synth_volume(0, 99, dims=dims).copy()
    idx = rng.choice(vol.size, size=600, replace=False)
    specials = np.array([np.inf, -np.inf, np.nan, 1e-41, -3e-42, 1.1754942e-38, 3.4e38, -3.4e38], np.float32)
    vol[idx] = specials[rng.integers(0, specials.size, idx.size)]
    with np.errstate(all="ignore"):
        for alpha, ww, wl in ((6.0, 1.0, 0.5), (1.5, 0.7, 0.45)):
            p = synth.brats_scene(0, 0, 120, dims=dims, image_hw=(72, 88), channels=1, intensity_alpha=alpha)
            p["ww"], p["wl"] = np.float32(ww), np.float32(wl)
            oext = dict(synth.SHADE_EXT) if shade else None
            ref, aux = oracle_c.brats_main(p, [vol], None, None, oext, return_aux=True)
            g = mrirt.upload_grid(vol, dims, layout)
            ext = dict(oext or {}, layout=layout)
            got, st = mrirt.render_brats(p, [g], ext=ext, stats=True)
            a, b = got.cpu().numpy(), ref
            assert np.array_equal(np.isnan(a), np.isnan(b)), "NaN pixels differ"
            assert np.array_equal(np.nan_to_num(a, nan=-1.0), np.nan_to_num(b, nan=-1.0)), float(np.nanmax(np.abs(a - b)))
            assert st["live_samples"] == aux["live_samples"]
            if layout in ("quad", "vg", "vga"):
                sk = mrirt.render_brats(p, [g], ext=ext, skip=True).cpu().numpy()
                assert np.array_equal(np.nan_to_num(sk, nan=-1.0), np.nan_to_num(a, nan=-1.0))
