"""Randomised and edge-case parity of K1 (STRICT) against the C oracle: the bit-faithfulness claim
has to hold for arbitrary cameras, grid shapes, clip planes, transfer functions and layouts, not only
for the bench scene."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-6


@pytest.fixture(scope="module")
def env():
    import torch
    import mrirt
    from mrirt import synth
    from oracle import oracle_c
    assert torch.cuda.is_available()
    return mrirt, synth, oracle_c


def _render_both(env, p, vols, lab, prd, ext, layout):
    mrirt, synth, oc = env
    dims = tuple(int(v) for v in p["dims"])
    okeys = ("cameraMode", "orthoHalfHeight", "shadeMode", "ka", "kd", "ks", "specPow2", "gradEps", "ertThreshold")
    ref, aux = oc.brats_main(p, vols, lab, prd, {k: v for k, v in (ext or {}).items() if k in okeys}, return_aux=True)
    lab_layout = "linear" if layout == "linear" else "brick"
    g = [mrirt.upload_grid(v, dims, layout) for v in vols]
    gl, gp = mrirt.upload_grid(lab, dims, lab_layout), mrirt.upload_grid(prd, dims, lab_layout)
    img, st = mrirt.render_brats(p, g, gl, gp, ext=dict(ext or {}, math="strict"), stats=True)
    return img.cpu().numpy(), st, ref, aux


@pytest.mark.parametrize("seed", range(16))
def test_random_scene(env, seed):
    mrirt, synth, oc = env
    rng = np.random.default_rng(1000 + seed)
    dims = tuple(int(v) for v in rng.integers(2, 40, 3))
    vols = [synth.synth_volume(0, seed * 7 + m, phase=float(rng.uniform(0, 3)), dims=dims) for m in range(4)]
    lab = (synth.synth_labels(0, dims=dims) + rng.integers(0, 2, dims[0] * dims[1] * dims[2]).astype(np.uint32) * 9) % 11
    lab = lab.astype(np.uint32)
    prd = np.roll(lab, 5).copy()
    cam = synth.bench_camera(radius=float(rng.uniform(0.3, 4.0)), phi_deg=float(rng.uniform(3, 177)),
                             theta_deg=float(rng.uniform(0, 360)))
    shade = bool(rng.integers(0, 2))
    p = synth.brats_scene(0, 0, int(rng.integers(6, 120)), dims=dims,
                          image_hw=(int(rng.integers(1, 60)), int(rng.integers(1, 60))),
                          channels=int(rng.integers(0, 5)), show_seg=bool(rng.integers(0, 2)),
                          show_pred=bool(rng.integers(0, 2)), intensity_alpha=float(rng.choice([0.4, 3.0, 16.0, 80.0])),
                          camera=cam, fov_deg=float(rng.uniform(15, 100)))
    p["nearT"], p["farT"] = float(rng.choice([0.0, 0.7, 2.0])), float(rng.choice([0.0, 2.5, 0.5]))
    p["gamma"] = float(rng.choice([1.0, 1.0, 0.6, 2.2]))
    p["ww"], p["wl"] = float(rng.choice([1.0, 0.5, 2.0])), float(rng.choice([0.5, 0.3]))
    p["bgColor"] = rng.random(3).astype(np.float32)
    p["volWeight"] = tuple(float(v) for v in rng.uniform(0.1, 3.0, 4))
    p["voxelSize"] = (p["voxelSize"] * rng.uniform(0.5, 2.0, 3)).astype(np.float32)     # anisotropic voxels
    ext = dict(synth.SHADE_EXT, cameraMode=int(rng.integers(0, 2)), orthoHalfHeight=float(rng.uniform(0.5, 1.5))) if shade else \
        dict(cameraMode=int(rng.integers(0, 2)), orthoHalfHeight=1.0)
    layouts = ["linear", "brick", "vg"] + ([] if shade else ["quad"])
    layout = layouts[int(rng.integers(0, len(layouts)))]
    got, st, ref, aux = _render_both(env, p, vols, lab, prd, ext, layout)
    assert np.abs(got - ref).max() <= TOL, (seed, layout, dims, np.abs(got - ref).max())
    assert st["live_samples"] == aux["live_samples"] and st["shaded_samples"] == aux["shaded_samples"]


def test_camera_inside_volume_and_axis_aligned_rays(env):
    mrirt, synth, oc = env
    dims = (16, 16, 16)
    vols = [synth.synth_volume(16, 3)]
    lab = synth.synth_labels(16)
    from mrirt.camera import OrbitalCamera
    # eye at the box centre, looking down an axis: tmin < 0, two direction components are exactly 0
    # at the centre pixel of an odd-sized image (the 1e-6 nudge of brats_rt.slang:96-98)
    cam = OrbitalCamera(initial_radius=0.0)
    p = synth.brats_scene(16, 0, 40, image_hw=(33, 33), channels=1, show_seg=True, intensity_alpha=3.0, camera=cam)
    p["eye"], p["U"], p["V"], p["W"] = (np.float32([0, 0, 0]), np.float32([1, 0, 0]), np.float32([0, 1, 0]),
                                         np.float32([0, 0, -1]))
    got, st, ref, aux = _render_both(env, p, vols * 4, lab, lab, None, "quad")
    assert np.abs(got - ref).max() <= TOL and st["live_samples"] == aux["live_samples"] > 0
    # orthographic, exactly axis-aligned: every ray has two zero direction components
    p2 = synth.brats_scene(16, 0, 40, image_hw=(24, 20), channels=1, intensity_alpha=3.0)
    p2["eye"], p2["U"], p2["V"], p2["W"] = (np.float32([0, 0, 3]), np.float32([1, 0, 0]), np.float32([0, 1, 0]),
                                            np.float32([0, 0, -1]))
    got, st, ref, aux = _render_both(env, p2, vols * 4, lab, lab, dict(cameraMode=1, orthoHalfHeight=1.0), "brick")
    assert np.abs(got - ref).max() <= TOL and st["live_samples"] == aux["live_samples"] > 0


def test_division_fallback_when_reciprocal_is_not_exact(env):
    """A voxelSize / ww / weight-sum whose significand is all ones is the one case Markstein's
    division excludes: the kernel must fall back to a true divide and stay bit-faithful."""
    mrirt, synth, oc = env
    dims = (12, 10, 8)
    vols = [synth.synth_volume(0, 5 + m, dims=dims) for m in range(4)]
    lab = synth.synth_labels(0, dims=dims)
    ones = np.frombuffer(np.uint32(0x3DFFFFFF).tobytes(), np.float32)[0]          # 0.12499999, significand all ones
    p = synth.brats_scene(0, 0, 48, dims=dims, image_hw=(30, 36), channels=2, intensity_alpha=6.0)
    p["voxelSize"] = np.float32([ones, 0.15, ones])
    p["volMin"] = (-0.5 * p["voxelSize"] * np.float32(dims)).astype(np.float32)
    p["ww"] = float(np.frombuffer(np.uint32(0x3F7FFFFF).tobytes(), np.float32)[0])      # 0.99999994
    p["volWeight"] = (float(np.frombuffer(np.uint32(0x3EFFFFFF).tobytes(), np.float32)[0]),) * 4   # sums to 0.99999994
    for layout in ("linear", "quad"):
        got, st, ref, aux = _render_both(env, p, vols, lab, lab, None, layout)
        assert np.abs(got - ref).max() <= TOL and st["live_samples"] == aux["live_samples"]


def test_no_modality_enabled_and_lut_edge_labels(env):
    mrirt, synth, oc = env
    dims = (10, 9, 8)
    vols = [synth.synth_volume(0, 1, dims=dims)] * 4
    lab = (np.arange(dims[0] * dims[1] * dims[2]) % 10).astype(np.uint32)      # labels 0..9: 8 and 9 are ignored (:145)
    p = synth.brats_scene(0, 0, 40, dims=dims, image_hw=(20, 28), channels=0, show_seg=True, show_pred=True)
    lut = np.array(p["lutColorAlpha"], np.float32)
    lut[5], lut[6], lut[7] = [0.2, 0.9, 0.4, 2.5], [0.7, 0.7, 0.1, 0.0], [0.3, 0.1, 0.8, 40.0]
    p["lutColorAlpha"] = [tuple(map(float, r)) for r in lut]
    got, st, ref, aux = _render_both(env, p, vols, lab, np.roll(lab, 3), None, "brick")
    assert np.abs(got - ref).max() <= TOL and st["live_samples"] == aux["live_samples"]
    assert ref[..., :3].max() > 0.1


@pytest.mark.parametrize("layout,shade", [("vg", True), ("quad", False)])
def test_exp_range_specialisation_both_sides_of_the_switch(env, layout, shade):
    """The strict exp drops its range reduction when intensityAlpha * stepSize <= 1/8 (a per-launch switch): frames
    just below, at and above the switch must all be the oracle's bits, in the pipelined kernels that use it."""
    mrirt, synth, oc = env
    n, image, steps = 40, 72, 96
    vol = synth.synth_volume(n)
    g = mrirt.upload_grid(vol, (n, n, n), layout)
    ext = dict(synth.SHADE_EXT) if shade else {}
    for target in (0.1249, 0.125, 0.1251, 0.9):
        p = synth.brats_scene(n, image, steps, channels=1, intensity_alpha=1.0)
        p["intensityAlpha"] = np.float32(target / float(p["stepSize"]))
        okeys = ("shadeMode", "ka", "kd", "ks", "specPow2", "gradEps")
        ref, aux = oc.brats_main(p, [vol], None, None, {k: v for k, v in ext.items() if k in okeys}, return_aux=True)
        got, st = mrirt.render_brats(p, [g], ext=dict(ext, layout=layout, math="strict"), stats=True)
        assert np.array_equal(got.cpu().numpy(), ref), target
        assert st["live_samples"] == aux["live_samples"]


@pytest.mark.parametrize("seed", range(12))
def test_k2_random_cameras_and_slabs(env, seed):
    """K2 walks only the steps between the (widened) slab interval of the cube and adds its way to it; whatever the
    camera, near / far planes, field of view and step count — camera inside the cube, segments that miss it, thousands
    of steps (the running sum drifts), a zero-length segment — the frame and the live-sample count are the oracle's."""
    mrirt, synth, oc = env
    rng = np.random.default_rng(4000 + seed)
    dims = tuple(int(v) for v in rng.integers(2, 36, 3))
    u8 = rng.integers(0, 256, size=dims[::-1], dtype=np.uint8)
    cam = synth.bench_camera(radius=float(rng.choice([0.2, 0.9, 1.6, 3.0, 6.0])), phi_deg=float(rng.uniform(2, 178)),
                             theta_deg=float(rng.uniform(0, 360)))
    near = float(rng.choice([0.0, 0.3, 1.5, 4.0]))
    far = near + float(rng.choice([0.0, 0.05, 1.0, 4.0, 12.0]))
    steps = int(rng.choice([1, 7, 64, 300, 3000]))
    p = synth.volume_scene(0, 0, steps, near, far, fov_deg=float(rng.uniform(20, 110)), camera=cam, dims=dims)
    p["imageSize"] = (np.uint32(int(rng.integers(1, 70))), np.uint32(int(rng.integers(1, 50))))
    ext = dict(cameraMode=int(rng.integers(0, 2)), orthoHalfHeight=float(rng.uniform(0.4, 1.6)))
    ref, aux = oc.volume_cs(p, u8, mode="u8", ext={k: ext[k] for k in ("cameraMode", "orthoHalfHeight")}, return_aux=True)
    img, st = mrirt.render_volume_u8(p, u8.reshape(-1), mode="u8", ext=ext, stats=True)
    assert np.array_equal(img.cpu().numpy(), ref), (seed, dims, steps, near, far)
    assert st["live_samples"] == aux["live_samples"]
    c8 = mrirt.render.build_cell8(u8.reshape(-1), dims, "u8")
    assert np.array_equal(mrirt.render_volume_u8(p, c8, mode="cell8", ext=ext).cpu().numpy(), ref)


@pytest.mark.parametrize("seed", range(12))
def test_vga_layout_and_slab_kernel_random_cameras(env, seed):
    """VGA (three axis-flat copies) through the register-gather kernels and through the LDS-staged slab kernel, against
    the C oracle: cameras around and inside the volume so that every copy (x-, y-, z-flat) and both march
    directions are read, windows that touch the volume's border, grids barely larger than a plane window,
    packets whose rays enter through different faces (those lanes take the global fall-back)."""
    mrirt, synth, oc = env
    rng = np.random.default_rng(7000 + seed)
    dims = tuple(int(v) for v in rng.integers(20, 56, 3))
    vol = synth.synth_volume(0, 50 + seed, phase=float(rng.uniform(0, 3)), dims=dims)
    ups = (None, (1.0, 0.0, 0.0), (0.0, 0.0, 1.0))
    cam = synth.bench_camera(radius=float(rng.choice([0.2, 0.8, 2.0, 3.5])), phi_deg=float(rng.uniform(3, 177)),
                             theta_deg=float(rng.uniform(0, 360)), world_up=None if seed % 3 == 0 else np.array(ups[seed % 3], np.float32))
    shade = bool(seed & 1)
    p = synth.brats_scene(0, 0, int(rng.integers(40, 260)), dims=dims, image_hw=(int(rng.integers(9, 90)), int(rng.integers(9, 90))),
                          channels=1, intensity_alpha=float(rng.choice([0.4, 16.0, 60.0])), camera=cam, fov_deg=float(rng.uniform(20, 95)))
    p["voxelSize"] = (p["voxelSize"] * rng.uniform(0.6, 1.7, 3)).astype(np.float32)
    p["gamma"] = float(rng.choice([1.0, 1.0, 1.8]))
    ext = dict(synth.SHADE_EXT) if shade else {}
    okeys = ("shadeMode", "ka", "kd", "ks", "specPow2", "gradEps")
    ref, aux = oc.brats_main(p, [vol], None, None, {k: v for k, v in ext.items() if k in okeys}, return_aux=True)
    g = mrirt.upload_grid(vol, dims, "vga")
    for variant in (0, 4, 64):                     # pipelined gathers, generic kernel, LDS-staged slab kernel
        got, st = mrirt.render_brats(p, [g], ext=dict(ext, math="strict", kernelVariant=variant), stats=True)
        assert np.array_equal(got.cpu().numpy(), ref), (seed, variant, dims, float(np.abs(got.cpu().numpy() - ref).max()))
        assert st["live_samples"] == aux["live_samples"] and st["shaded_samples"] == aux["shaded_samples"]


@pytest.mark.parametrize("seed", range(8))
def test_vga_modalities_overlays_and_workgroup_shapes(env, seed):
    """VGA grids beyond the single shaded modality: 1-4 enabled modalities with weights (the pipelined, the rolling and
    the generic kernels), seg / prediction overlays, both workgroup shapes (kernelVariant 0: 16 x 16 pixels, 2: one
    8 x 8 packet), image sides that are not multiples of either, strict math — bit for bit against the C oracle,
    counters included."""
    mrirt, synth, oc = env
    rng = np.random.default_rng(9100 + seed)
    dims = tuple(int(v) for v in rng.integers(18, 44, 3))
    nmod = 1 + seed % 4
    vols = [synth.synth_volume(0, 80 + 7 * seed + m, phase=0.4 * m, dims=dims) for m in range(4)]
    lab = synth.synth_labels(0, dims=dims)
    pred = np.roll(lab.reshape(dims[::-1]), 3, axis=2).reshape(-1).copy()
    cam = synth.bench_camera(radius=float(rng.choice([0.3, 1.2, 2.6])), phi_deg=float(rng.uniform(10, 170)), theta_deg=float(rng.uniform(0, 360)))
    show_seg, show_pred = bool(seed & 2), bool(seed & 4)
    p = synth.brats_scene(0, 0, int(rng.integers(50, 200)), dims=dims, image_hw=(int(rng.integers(11, 70)), int(rng.integers(11, 70))),
                          channels=nmod, show_seg=show_seg, show_pred=show_pred, intensity_alpha=float(rng.choice([0.8, 12.0])),
                          camera=cam, fov_deg=float(rng.uniform(25, 80)))
    p["volWeight"] = tuple(float(v) for v in rng.uniform(0.3, 1.4, 4))
    okeys = ("shadeMode", "ka", "kd", "ks", "specPow2", "gradEps")
    grids = [mrirt.upload_grid(v, dims, "vga") for v in vols]
    gl, gp = mrirt.upload_grid(lab, dims, "brick"), mrirt.upload_grid(pred, dims, "brick")
    for shade in (True, False):
        ext = dict(synth.SHADE_EXT) if shade else {}
        ref, aux = oc.brats_main(p, vols, lab if show_seg else None, pred if show_pred else None,
                                 {k: v for k, v in ext.items() if k in okeys}, return_aux=True)
        for variant in (0, 2):
            got, st = mrirt.render_brats(p, grids, labels=gl if show_seg else None, preds=gp if show_pred else None,
                                         ext=dict(ext, layout="vga", kernelVariant=variant), stats=True)
            assert np.array_equal(got.cpu().numpy(), ref), (seed, shade, variant, float(np.abs(got.cpu().numpy() - ref).max()))
            assert st["live_samples"] == aux["live_samples"] and st["shaded_samples"] == aux["shaded_samples"]


def test_slab_kernel_refuses_grids_smaller_than_a_window(env):
    mrirt, synth, oc = env
    dims = (12, 30, 30)
    vol = synth.synth_volume(0, 3, dims=dims)
    p = synth.brats_scene(0, 0, 64, dims=dims, image_hw=(16, 16), channels=1)
    g = mrirt.upload_grid(vol, dims, "vga")
    with pytest.raises(mrirt._lib.MrirtError):
        mrirt.render_brats(p, [g], ext=dict(kernelVariant=64))
    ref = oc.brats_main(p, [vol], None, None)
    assert np.array_equal(mrirt.render_brats(p, [g]).cpu().numpy(), ref)
