"""HIP-vs-oracle parity on a real MI355X, through the C ABI (libmrirt.so via ctypes).

Bars (BASELINE.json: images within 1e-4 max-abs of the CPU reference, fp32):
  * STRICT math: the kernels evaluate the oracle's unfused fp32 expression tree with fp64-backed
    exp, so the images are compared at 1e-6 max-abs and the live-sample counts must be equal.
  * FAST math (FMA + hardware exp2/rcp): 1e-4 max-abs over robust pixels; pixels the oracle
    flags as knife-edge for early termination (T within 1e-5 of 0.01, t within 4 ulp of t1) are
    counted, must be rare, and are bounded by the analytic flip error ert*max(val, lut.rgb).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STRICT_TOL = 1e-6
FAST_TOL = 1e-4


@pytest.fixture(scope="module")
def env():
    import torch
    import mrirt
    from mrirt import synth
    from oracle import oracle_c, oracle_np
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName
    mrirt._lib.lib()          # raises if libmrirt.so is missing: no fallback
    return dict(torch=torch, mrirt=mrirt, synth=synth, oc=oracle_c, onp=oracle_np)


@pytest.fixture(scope="module")
def scene48(env):
    synth = env["synth"]
    dims = (48, 40, 36)
    vols = [synth.synth_volume(0, 1234 + m, phase=0.3 * m, dims=dims) for m in range(4)]
    lab = synth.synth_labels(0, dims=dims)
    prd = np.roll(lab, 7).copy()
    return dims, vols, lab, prd


K1_CASES = {
    "viewer_default_4ch_seg": (dict(channels=4, show_seg=True), None),
    "one_channel": (dict(channels=1), None),
    "dense_seg_pred": (dict(channels=2, show_seg=True, show_pred=True, intensity_alpha=16.0), None),
    "shade_dense": (dict(channels=1, intensity_alpha=16.0), "SHADE"),
    "shade_3ch_ortho": (dict(channels=3, intensity_alpha=4.0), "SHADE_ORTHO"),
}


def _ext(env, tag):
    synth = env["synth"]
    if tag is None:
        return None
    if tag == "SHADE":
        return dict(synth.SHADE_EXT)
    return dict(synth.SHADE_EXT, cameraMode=1, orthoHalfHeight=1.1)


def _oracle_ext(ext):
    keys = ("cameraMode", "orthoHalfHeight", "shadeMode", "ka", "kd", "ks", "specPow2", "gradEps", "ertThreshold")
    return {k: v for k, v in (ext or {}).items() if k in keys and v is not None}


@pytest.mark.parametrize("case", sorted(K1_CASES))
@pytest.mark.parametrize("layout", ["linear", "brick", "vg", "quad"])
def test_k1_strict_matches_oracle(env, scene48, case, layout):
    mrirt, synth, oc = env["mrirt"], env["synth"], env["oc"]
    dims, vols, lab, prd = scene48
    kw, tag = K1_CASES[case]
    ext = _ext(env, tag)
    p = synth.brats_scene(0, 0, 96, dims=dims, image_hw=(72, 88), **kw)
    ref, aux = oc.brats_main(p, vols, lab, prd, _oracle_ext(ext), return_aux=True)
    g = [mrirt.upload_grid(v, dims, layout) for v in vols]
    lab_layout = "linear" if layout == "linear" else "brick"
    gl, gp = mrirt.upload_grid(lab, dims, lab_layout), mrirt.upload_grid(prd, dims, lab_layout)
    if layout == "quad" and tag is not None:
        with pytest.raises(mrirt._lib.MrirtError):     # QUAD grids carry no gradient neighbours
            mrirt.render_brats(p, g, gl, gp, ext=dict(ext, math="strict"))
        return
    img, st = mrirt.render_brats(p, g, gl, gp, ext=dict(ext or {}, math="strict"), stats=True)
    got = img.cpu().numpy()
    assert got.shape == ref.shape and got.dtype == np.float32
    err = np.abs(got - ref).max()
    assert err <= STRICT_TOL, f"{case}/{layout}: max-abs {err:.3e}"
    assert st["live_samples"] == aux["live_samples"]
    assert st["shaded_samples"] == aux["shaded_samples"]
    assert np.all(got[..., 3] == 1.0)


def _fast_check(got, ref, aux, what):
    diff = np.abs(got - ref)[..., :3].max(axis=-1)
    robust = ~aux["fragile"]
    assert diff[robust].max() <= FAST_TOL, f"{what}: robust max-abs {diff[robust].max():.3e}"
    assert aux["fragile"].mean() < 0.01
    # knife-edge pixels: one step more or less changes C by at most T*alpha*emission <= 0.01
    assert diff.max() <= 0.01 + FAST_TOL, f"{what}: max-abs {diff.max():.3e}"


# FAST math is the one mode whose arithmetic differs from the oracle's (FMA lerps, v_exp_f32, v_rcp_f32), so EVERY fast kernel
# family is held to the oracle here, not to itself: the generic kernel (linear / brick), the pipelined
# kernel (vg / vga: one modality; quad: 1-4 modalities), the rolling kernel (vg / vga with 2-4 shaded modalities) and the
# skipping march on top of the pipelined ones.
FAST_LAYOUTS = ["linear", "brick", "vg", "vga", "quad"]


@pytest.mark.parametrize("case", sorted(K1_CASES))
@pytest.mark.parametrize("layout", FAST_LAYOUTS)
def test_k1_fast_within_tolerance(env, scene48, case, layout):
    mrirt, synth, onp = env["mrirt"], env["synth"], env["onp"]
    dims, vols, lab, prd = scene48
    kw, tag = K1_CASES[case]
    ext = _ext(env, tag)
    if layout == "quad" and tag is not None:
        pytest.skip("QUAD grids carry no gradients")
    p = synth.brats_scene(0, 0, 96, dims=dims, image_hw=(72, 88), **kw)
    ref, aux = onp.brats_main(p, vols, lab, prd, _oracle_ext(ext), return_aux=True)
    g = [mrirt.upload_grid(v, dims, layout) for v in vols]
    lab_layout = "linear" if layout == "linear" else "brick"
    gl, gp = mrirt.upload_grid(lab, dims, lab_layout), mrirt.upload_grid(prd, dims, lab_layout)
    fx = dict(ext or {}, math="fast", layout=layout)
    got = mrirt.render_brats(p, g, gl, gp, ext=fx).cpu().numpy()
    _fast_check(got, ref, aux, f"{case}/{layout}")
    if layout in ("vg", "vga", "quad"):
        # the skipping march (where this configuration has one) must give the FAST kernel's own bits, hence the oracle's
        sk = mrirt.render_brats(p, g, gl, gp, ext=fx, skip=True).cpu().numpy()
        assert np.array_equal(sk, got), f"{case}/{layout}: skip changes the FAST frame"


@pytest.mark.parametrize("layout", ["vg", "vga", "quad"])
@pytest.mark.parametrize("channels", [1, 2, 4])
def test_k1_fast_skipping_and_rolling_kernels_match_oracle(env, layout, channels):
    """FAST math on a volume WITH empty space (so the skipping march really leaps), overlays off (so vg / vga with 2-4
    modalities takes the rolling kernel), shaded where the layout allows it — against oracle_np, plain and skip=True."""
    mrirt, synth, onp = env["mrirt"], env["synth"], env["onp"]
    from test_gpu_skip import head_in_air
    n = 56
    vols, lab = head_in_air(n, seed=21, channels=channels)
    shade = layout != "quad"
    p = synth.brats_scene(n, 112, 144, channels=channels, intensity_alpha=6.0)
    p["wl"], p["ww"] = np.float32(0.45), np.float32(0.7)
    ext = dict(synth.SHADE_EXT) if shade else {}
    ref, aux = onp.brats_main(p, vols, None, None, _oracle_ext(ext), return_aux=True)
    g = [mrirt.upload_grid(v, (n, n, n), layout) for v in vols]
    fx = dict(ext, math="fast", layout=layout)
    for skip in (False, True):
        got, st = mrirt.render_brats(p, g, ext=fx, stats=True, skip=skip)
        _fast_check(got.cpu().numpy(), ref, aux, f"{layout}/{channels}ch/skip={skip}")
        # counts may differ from the oracle's only on knife-edge rays (one step more or less each)
        assert abs(st["live_samples"] - aux["live_samples"]) <= int(aux["fragile"].sum()) + 2


def test_k1_plain_abi_entry(env, scene48):
    """mrirt_render_brats (no ext): the reference's layout and behaviour, raw pointers."""
    import ctypes as C
    torch, mrirt, synth, oc = env["torch"], env["mrirt"], env["synth"], env["oc"]
    from mrirt import _lib
    from mrirt.params import brats_params
    dims, vols, lab, prd = scene48
    p = synth.brats_scene(0, 0, 64, dims=dims, image_hw=(50, 70), channels=4, show_seg=True)
    ref = oc.brats_main(p, vols, lab, prd)
    dv = [torch.from_numpy(v).cuda() for v in vols]
    dl = torch.from_numpy(lab.view(np.int32)).cuda()
    out = torch.full((50, 80, 4), -1.0, device="cuda")      # pitch 80 > width 70
    P = brats_params(p)
    vp = (C.c_void_p * 4)(*[C.c_void_p(t.data_ptr()) for t in dv])
    rc = _lib.lib().mrirt_render_brats(C.byref(P), vp, C.c_void_p(dl.data_ptr()), None, C.c_void_p(out.data_ptr()),
                                       80, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    got = out.cpu().numpy()
    assert np.abs(got[:, :70] - ref).max() <= STRICT_TOL
    assert np.all(got[:, 70:] == -1.0), "pixels beyond imageSize must not be written"


def test_k1_error_codes(env, scene48):
    mrirt, synth = env["mrirt"], env["synth"]
    dims, vols, lab, prd = scene48
    p = synth.brats_scene(0, 0, 64, dims=dims, image_hw=(32, 32), channels=1)
    with pytest.raises(ValueError):
        mrirt.render_brats(p, [vols[0][:100]])                      # too small
    bad = dict(p, dims=np.array([1, 40, 36], np.uint32))
    with pytest.raises(Exception):
        mrirt.render_brats(bad, [vols[0]])
    with pytest.raises(KeyError):
        mrirt.render_brats({k: v for k, v in p.items() if k != "ww"}, [vols[0]])


def test_k1_half_output_and_ert_override(env, scene48):
    mrirt, synth, oc = env["mrirt"], env["synth"], env["oc"]
    dims, vols, lab, prd = scene48
    p = synth.brats_scene(0, 0, 96, dims=dims, image_hw=(40, 56), channels=1, intensity_alpha=16.0)
    ref = oc.brats_main(p, vols, None, None)
    h = mrirt.render_brats(p, [vols[0]], ext=dict(outFormat="rgba16f")).cpu().numpy()
    assert h.dtype == np.float16
    assert np.array_equal(h, ref.astype(np.float16))
    # ERT off (threshold 0): semantics differ from the reference by design, oracle agrees
    ref0 = oc.brats_main(p, vols, None, None, dict(ertThreshold=0.0))
    got0 = mrirt.render_brats(p, [vols[0]], ext=dict(ertThreshold=0.0)).cpu().numpy()
    assert np.abs(got0 - ref0).max() <= STRICT_TOL
    assert np.abs(ref0 - ref).max() > 1e-4          # the 0.01 cut-off is part of the semantics


def test_brick_roundtrip(env):
    torch, mrirt = env["torch"], env["mrirt"]
    rng = np.random.default_rng(5)
    for dims in ((8, 8, 4), (9, 7, 5), (33, 18, 3), (64, 64, 64)):
        for dt in (np.float32, np.uint8, np.int32):
            a = (rng.random(dims[0] * dims[1] * dims[2]) * 200).astype(dt)
            g = mrirt.upload_grid(a, dims, "brick")
            assert g.data.numel() == mrirt.render.brick_elems(dims)
            back = mrirt.unbrick_grid(g).cpu().numpy()
            assert np.array_equal(back, a), (dims, dt)
    # layout definition: voxel (x,y,z) sits at brick (x>>2,y>>2,z>>1), offset (x&3)+4(y&3)+16(z&1)
    dims = (9, 7, 5)
    a = np.arange(9 * 7 * 5, dtype=np.float32)
    b = mrirt.upload_grid(a, dims, "brick").data.cpu().numpy()
    nbx, nby = 3, 2
    for (x, y, z) in ((0, 0, 0), (8, 6, 4), (5, 3, 2), (4, 4, 1)):
        off = ((z >> 1) * nby + (y >> 2)) * nbx * 32 + (x >> 2) * 32 + (x & 3) + 4 * (y & 3) + 16 * (z & 1)
        assert b[off] == a[x + y * 9 + z * 63]


@pytest.mark.parametrize("mode", ["u32x4", "u8", "f32"])
def test_k2_matches_oracle(env, mode):
    mrirt, synth, oc, onp = env["mrirt"], env["synth"], env["oc"], env["onp"]
    dims = (40, 36, 30)
    f = synth.synth_volume(0, 1234, dims=dims)
    u8 = np.rint(f * 255).astype(np.uint8)
    vol = {"u32x4": onp.pack_u8_volume(u8), "u8": u8, "f32": f}[mode]
    for steps, near, far in ((64, 1.5, 4.5), (33, 2.6, 2.9), (16, 0.0, 9.0)):
        p = synth.volume_scene(0, 0, steps, near, far, dims=dims)
        p["imageSize"] = (np.uint32(76), np.uint32(52))
        ref, aux = oc.volume_cs(p, vol, mode=mode, return_aux=True)
        img, st = mrirt.render_volume_u8(p, vol, mode=mode, stats=True)
        got = img.cpu().numpy()
        assert np.array_equal(got, ref), f"{mode} steps {steps}: max-abs {np.abs(got - ref).max():.3e}"
        assert st["live_samples"] == aux["live_samples"]
        fast = mrirt.render_volume_u8(p, vol, mode=mode, ext=dict(math="fast")).cpu().numpy()
        assert np.abs(fast - ref).max() <= FAST_TOL + 0.005 * (ref.max() > 0.995)   # accum>0.995 break flips


def test_k2_ortho_f32_config1(env):
    """BASELINE config 1 at reduced size: K2 loop on an fp32 grid, orthographic segments."""
    mrirt, synth, oc = env["mrirt"], env["synth"], env["oc"]
    f = synth.synth_volume(32, 1234)
    p = synth.volume_scene(32, 64, 64)
    ext = dict(cameraMode=1, orthoHalfHeight=1.1)
    ref = oc.volume_cs(p, f, mode="f32", ext=ext)
    got = mrirt.render_volume_u8(p, f, mode="f32", ext=ext).cpu().numpy()
    assert np.array_equal(got, ref)


def test_k3_matches_oracle(env):
    mrirt, synth, oc = env["mrirt"], env["synth"], env["oc"]
    sp, eye, U, V, W = synth.sdf_scene()
    for (w, h) in ((64, 48), (80, 80), (33, 17)):
        ref = oc.raymarch_cs(sp, eye, U, V, W, w, h)
        got = mrirt.render_sdf(sp, eye, U, V, W, w, h).cpu().numpy()
        # hit colour goes through atan2 (fp64 on both sides); everything else is unfused fp32
        assert np.abs(got - ref).max() <= STRICT_TOL


def test_tiles_compact_and_detile(env, scene48):
    """Tile sharding on ONE device: N ranks' compact buffers, stacked, de-tile to the full frame."""
    torch, mrirt, synth = env["torch"], env["mrirt"], env["synth"]
    from mrirt import tiles
    dims, vols, lab, prd = scene48
    p = synth.brats_scene(0, 0, 64, dims=dims, image_hw=(100, 150), channels=1, intensity_alpha=16.0)
    g = [mrirt.upload_grid(vols[0], dims, "brick")]
    full = mrirt.render_brats(p, g).cpu().numpy()
    for world in (1, 2, 3, 8):
        parts = []
        max_local = tiles.local_tile_count(150, 100, 32, 0, world)
        for r in range(world):
            t = mrirt.render_brats(p, g, ext=tiles.shard_ext(None, r, world, 32))
            assert t.shape[0] == tiles.local_tile_count(150, 100, 32, r, world)
            if t.shape[0] < max_local:
                t = torch.cat([t, t.new_zeros((max_local - t.shape[0],) + tuple(t.shape[1:]))])
            parts.append(t)
        frame = tiles.assemble_frame(torch.stack(parts), 150, 100, 32, world).cpu().numpy()
        assert np.array_equal(frame, full), f"world {world}"


def test_shim_dispatch_like_the_viewer(env, scene48):
    """The reference's call shape: create_buffer/copy_from_numpy, create_texture(rgba16_float),
    kernel.dispatch(thread_count, vars={... gParams dict ...}, command_encoder)."""
    mrirt, synth, oc = env["mrirt"], env["synth"], env["oc"]
    from mrirt import shim
    dims, vols, lab, prd = scene48
    dev = shim.Device(enable_debug_layers=True)
    kernel = dev.create_compute_kernel(dev.load_program("brats_rt.slang", ["brats_main"]))
    bufs = []
    for v in vols:
        b = dev.create_buffer(element_count=v.size, struct_size=4, usage=shim.BufferUsage.shader_resource)
        b.copy_from_numpy(v)
        bufs.append(b)
    seg = dev.create_buffer(element_count=lab.size, struct_size=4)
    seg.copy_from_numpy(lab)
    empty_u = dev.create_buffer(element_count=1, struct_size=4)
    empty_u.copy_from_numpy(np.zeros(1, np.uint32))
    tex = dev.create_texture(format=shim.Format.rgba16_float, width=96, height=64,
                             usage=shim.TextureUsage.shader_resource | shim.TextureUsage.unordered_access)
    p = synth.brats_scene(0, 0, 96, dims=dims, image_hw=(64, 96), channels=4, show_seg=True)
    ce = dev.create_command_encoder()
    kernel.dispatch(thread_count=[96, 64, 1],
                    vars={"gOutput": tex, "gIntensity0": bufs[0], "gIntensity1": bufs[1], "gIntensity2": bufs[2],
                          "gIntensity3": bufs[3], "gLabels": seg, "gPreds": empty_u, "gParams": p},
                    command_encoder=ce)
    dev.submit_command_buffer(ce.finish())
    ref = oc.brats_main(p, vols, lab, None)
    assert np.array_equal(tex.to_numpy(), ref.astype(np.float16))
    with pytest.raises(KeyError):
        kernel.dispatch(thread_count=[96, 64, 1], vars={"gOutput": tex, "gParams": p}, command_encoder=ce)
    with pytest.raises(RuntimeError):
        dev.load_program("x.slang", ["no_such_entry"])


def test_full_size_properties(env):
    """BASELINE-sized run (512^3, 1024^2, 512 steps, gradient shading + ERT) checked through
    size-independent properties, plus an oracle comparison on a band of rows."""
    torch, mrirt, synth, oc = env["torch"], env["mrirt"], env["synth"], env["oc"]
    n, image, steps = 512, 1024, 512
    vol = synth.synth_volume(n)
    p = synth.brats_scene(n, image, steps, channels=1, intensity_alpha=16.0)
    ext = dict(synth.SHADE_EXT)
    gl = mrirt.upload_grid(vol, (n, n, n), "linear")
    gb = mrirt.upload_grid(vol, (n, n, n), "brick")
    a, sa = mrirt.render_brats(p, [gl], ext=ext, stats=True)
    b, sb = mrirt.render_brats(p, [gb], ext=ext, stats=True)
    assert torch.equal(a, b), "layout must not change a single bit"
    assert sa == sb and 0 < sa["live_samples"] < image * image * steps
    gv = mrirt.upload_grid(vol, (n, n, n), "vg")
    c, sc = mrirt.render_brats(p, [gv], ext=ext, stats=True)
    assert torch.equal(a, c) and sc == sa, "value+gradient grid must not change a single bit"
    # unshaded: QUAD and VG grids agree with the linear grid bit for bit as well
    gq = mrirt.upload_grid(vol, (n, n, n), "quad")
    u0 = mrirt.render_brats(p, [gl])
    assert torch.equal(u0, mrirt.render_brats(p, [gq])) and torch.equal(u0, mrirt.render_brats(p, [gv]))
    # the three axis-flat copies (VGA): register-gather kernels and the LDS-staged slab kernel (kernelVariant 64)
    ga = mrirt.upload_grid(vol, (n, n, n), "vga")
    for variant in (0, 64):
        d, sd = mrirt.render_brats(p, [ga], ext=dict(ext, kernelVariant=variant), stats=True)
        assert torch.equal(a, d) and sd == sa, f"VGA grid, kernel variant {variant}: must not change a single bit"
        assert torch.equal(u0, mrirt.render_brats(p, [ga], ext=dict(kernelVariant=variant)))
    del gq, gv, ga, u0, c, d
    img = b.cpu().numpy()
    assert np.isfinite(img).all() and img[..., :3].min() >= 0.0 and np.all(img[..., 3] == 1.0)
    assert np.array_equal(img[..., 0], img[..., 1]) and np.array_equal(img[..., 0], img[..., 2])   # grey emission
    # idempotence / determinism
    assert torch.equal(b, mrirt.render_brats(p, [gb], ext=ext))
    # tile-sharded rendering reproduces the frame bit for bit
    from mrirt import tiles
    parts = [mrirt.render_brats(p, [gb], ext=tiles.shard_ext(ext, r, 4, 64)) for r in range(4)]
    assert torch.equal(tiles.assemble_frame(torch.stack(parts), image, image, 64, 4), b)
    # the oracle over the WHOLE frame — all 1024 rows (the C/OpenMP oracle renders config 3 in a couple of
    # seconds on the GPU box's host): image bits and the live-sample count
    ref, aux = oc.brats_main(p, [vol], None, None, synth.SHADE_EXT, return_aux=True)
    assert ref.shape == img.shape
    assert np.abs(img - ref).max() <= STRICT_TOL and np.array_equal(img, ref)
    assert aux["live_samples"] == sa["live_samples"]
    # ERT really fires in this configuration, and linearity in bgColor holds:
    # C(bg) = C(0) + bg exactly where no sample contributes; elsewhere C >= bg
    p2 = dict(p, bgColor=np.array([0.25, 0.25, 0.25], np.float32))
    c2 = mrirt.render_brats(p2, [gb], ext=ext).cpu().numpy()
    miss = img[..., 0] == 0.0
    assert np.all(c2[..., 0][miss] == 0.25)


def test_bc4_decode_on_device_matches_host_and_oracle():
    """mrirt_bc4_decode == volume.bc4_decode == the oracle's decode, incl. ragged (non-multiple-of-4) sizes
    and both palette modes (r0 > r1 and r0 <= r1)."""
    import torch
    from mrirt import volume
    from oracle import oracle_np as onp
    rng = np.random.default_rng(11)
    for (w, h, d) in ((16, 12, 5), (13, 7, 3), (4, 4, 1), (1, 1, 2), (37, 41, 9)):
        bw, bh = (w + 3) // 4, (h + 3) // 4
        blk = rng.integers(0, 256, size=(d, bh, bw, 8), dtype=np.uint8)
        blk[0, 0, 0, :2] = (200, 10)          # six-interpolant mode
        blk[-1, -1, -1, :2] = (10, 200)       # four-interpolant mode with 0 / 255
        if bw > 1:
            blk[0, 0, 1, :2] = (77, 77)       # r0 == r1 -> four-interpolant branch
        data = blk.tobytes()
        host = volume.bc4_decode(data, w, h, d)
        assert np.array_equal(host, onp.bc4_decode(data, w, h, d))
        dev = volume.bc4_decode_device(data, w, h, d)
        assert dev.dtype == torch.uint8 and dev.numel() == w * h * d
        assert np.array_equal(dev.cpu().numpy(), host)


@pytest.mark.parametrize("math", ["strict", "fast"])
def test_k2_cell8_layout_is_the_same_frame(math):
    """MRIRT_VOX_CELL8 (one 8-byte gather per sample) against the byte and u32-per-voxel modes and the oracle,
    on a ragged volume whose last voxels exercise the p1 = min(p0 + 1, d - 1) clamp the build kernel repeats."""
    import torch
    import mrirt
    from mrirt import synth, volume
    from oracle import oracle_c
    dims = (37, 29, 23)
    rng = np.random.default_rng(21)
    u8 = rng.integers(0, 256, size=dims[::-1], dtype=np.uint8)          # every byte value occurs
    pk = volume.pack_u8_as_u32x4(u8)
    p = synth.volume_scene(0, 96, 80, 1.0, 5.0, dims=dims)
    ext = dict(math=math)
    t8 = torch.from_numpy(u8.reshape(-1)).cuda()
    a = mrirt.render_volume_u8(p, t8, mode="u8", ext=ext)
    for src, mode in ((u8.reshape(-1), "u8"), (pk.view(np.int32).reshape(-1), "u32x4")):
        c8 = mrirt.render.build_cell8(src, dims, mode)
        b, st = mrirt.render_volume_u8(p, c8, mode="cell8", ext=ext, stats=True)
        assert torch.equal(a, b)
        assert st["live_samples"] > 0
    if math == "strict":
        assert np.array_equal(a.cpu().numpy(), oracle_c.volume_cs(p, pk, mode="u32x4"))


def test_shim_k2_marches_the_cell8_copy():
    """The slangpy-shaped dispatch of volume_cs: same frame whether the Device converts (default) or not."""
    import mrirt.shim as spy
    from mrirt import synth, volume
    dims = (40, 36, 28)
    u8 = np.random.default_rng(2).integers(0, 256, size=dims[::-1], dtype=np.uint8)
    pk = volume.pack_u8_as_u32x4(u8)
    p = synth.volume_scene(0, 64, 48, 1.0, 5.0, dims=dims)
    frames = []
    for layout in ("auto", "linear"):
        dev = spy.Device(layout=layout)
        k = dev.create_compute_kernel(dev.load_program("volume_render.slang", ["volume_cs"]))
        buf = dev.create_buffer(element_count=pk.size // 4, struct_size=16)
        buf.copy_from_numpy(pk)
        tex = dev.create_texture(format=spy.Format.rgba32_float, width=64, height=64)
        k.dispatch(thread_count=[64, 64, 1], vars={"gOutput": tex, "gParams": p, "gVolumeU8": buf})
        frames.append(tex.to_numpy())
    assert np.array_equal(frames[0], frames[1]) and frames[0][..., 0].max() > 0.05


@pytest.mark.parametrize("layout", ["vg", "vga"])          # vga (16 x 16-pixel workgroups) is what bench.py --gpus N marches
def test_c4_full_size_tiles_over_eight_ranks(env, layout):
    """BASELINE config 4 at full size on one GPU: 512^3, 2048 x 2048, 512 steps, shaded, 64 x 64 tiles dealt to 8
    'ranks'.  Each rank's compact tile buffer, gathered and de-tiled (the HIP de-tiling kernel), must be the
    whole-frame render bit for bit, and the ranks' live-sample counts must add up to the whole frame's."""
    torch, mrirt, synth = env["torch"], env["mrirt"], env["synth"]
    from mrirt import tiles
    n, image, steps, world, tile = 512, 2048, 512, 8, 64
    vol = synth.synth_volume(n)
    g = mrirt.upload_grid(vol, (n, n, n), layout)
    p = synth.brats_scene(n, image, steps, channels=1, intensity_alpha=16.0)
    ext = dict(synth.SHADE_EXT, layout=layout)
    whole, sw = mrirt.render_brats(p, [g], ext=ext, stats=True)
    max_local = tiles.local_tile_count(image, image, tile, 0, world)
    # the deal bench.py --gpus N uses (tiles along diagonals: tiles.balanced_skew), and the plain row-major one
    for skew in (tiles.balanced_skew(image, tile, world), 0):
        gathered = torch.zeros((world, max_local, tile, tile, 4), device="cuda")
        live = shaded = 0
        per_rank = []
        for r in range(world):
            part, st = mrirt.render_brats(p, [g], ext=tiles.shard_ext(ext, r, world, tile, skew), stats=True)
            assert part.shape == (tiles.local_tile_count(image, image, tile, r, world), tile, tile, 4)
            gathered[r, :part.shape[0]] = part
            live, shaded = live + st["live_samples"], shaded + st["shaded_samples"]
            per_rank.append(st["live_samples"])
        assert torch.equal(mrirt.detile(gathered, image, image, tile, world, skew=skew), whole), skew
        assert live == sw["live_samples"] and shaded == sw["shaded_samples"]
        if skew:          # the diagonal deal balances the ranks' work: within 3 % of an eighth of the frame each
            assert max(per_rank) <= 1.03 * live / world, per_rank
    assert 0 < live < image * image * steps
    # ... and the frame itself against the oracle: 128 rows through the long centre rays and 32 at the top edge of the
    # 2048^2 image (C4 used to be compared only with itself)
    oc = env["oc"]
    img = whole.cpu().numpy()
    for rows in ((960, 1088), (0, 32)):
        ref = oc.brats_main(p, [vol], None, None, synth.SHADE_EXT, rows=rows)
        assert np.array_equal(img[rows[0]:rows[1]], ref), rows


def test_c2_full_size_against_the_oracle(env):
    """BASELINE config 2 at full size, the reference's own semantics: 256^3 x 4 modalities + segmentation overlay,
    512 x 512 perspective, 256 steps.  The C/OpenMP oracle renders the whole frame in a fraction of a second on the
    GPU box's host, so this one is compared in full: image bits and live-sample count, for the float4 layout the
    shim picks and for the reference's own linear buffers."""
    torch, mrirt, synth, oc = env["torch"], env["mrirt"], env["synth"], env["oc"]
    n, image, steps = 256, 512, 256
    vols = [synth.synth_volume(n, 1234 + m, phase=0.4 * m) for m in range(4)]
    lab = synth.synth_labels(n)
    p = synth.brats_scene(n, image, steps, channels=4, show_seg=True, intensity_alpha=0.4)
    ref, aux = oc.brats_main(p, vols, lab, None, None, return_aux=True)
    for layout, lab_layout in (("quad", "brick"), ("quad", "labcell"), ("linear", "linear"), ("mod4", "brick"), ("mod4", "labcell")):
        # ("mod4": the four modalities as ONE float4 grid, MRIRT_LAYOUT_MOD4 — a quarter of the four quad grids' memory)
        g = [mrirt.upload_mod4(vols, (n, n, n))] * 4 if layout == "mod4" else [mrirt.upload_grid(v, (n, n, n), layout) for v in vols]
        # ("labcell": the seg overlay as label cells — what bench.py's k1_reference_path and the shim bind on QUAD grids)
        gl = mrirt.upload_label_cells(lab, None, (n, n, n)) if lab_layout == "labcell" else mrirt.upload_grid(lab, (n, n, n), lab_layout)
        got, st = mrirt.render_brats(p, g, gl, ext=dict(layout=layout, labelLayout=lab_layout), stats=True, skip=layout == "quad" and lab_layout == "brick")
        assert np.array_equal(got.cpu().numpy(), ref), layout
        assert st["live_samples"] == aux["live_samples"]
        if layout == "mod4" and lab_layout == "brick":
            # which modalities are enabled is a run-time property of the one MOD4 kernel: every subset against four quad grids
            gq = [mrirt.upload_grid(v, (n, n, n), "quad") for v in vols]
            for en in ((1, 0, 0, 0), (0, 1, 1, 0), (0, 0, 0, 1), (1, 1, 0, 1), (0, 0, 0, 0)):
                pe = dict(p, volEnabled=en, imageSize=(160, 128))
                a4 = mrirt.render_brats(pe, g, gl, ext=dict(layout="mod4", labelLayout="brick"))
                aq = mrirt.render_brats(pe, gq, gl, ext=dict(layout="quad", labelLayout="brick"))
                assert torch.equal(a4, aq), en
            pf = dict(p, gamma=1.7, imageSize=(160, 128))            # the general-gamma and the FAST kernels
            assert torch.equal(mrirt.render_brats(pf, g, gl, ext=dict(layout="mod4", labelLayout="brick")),
                               mrirt.render_brats(pf, gq, gl, ext=dict(layout="quad", labelLayout="brick")))
            fa = mrirt.render_brats(pf, g, gl, ext=dict(layout="mod4", labelLayout="brick", math="fast"))
            fq = mrirt.render_brats(pf, gq, gl, ext=dict(layout="quad", labelLayout="brick", math="fast"))
            assert float((fa - fq).abs().max()) < 1e-5
            with pytest.raises(RuntimeError):                        # no gradients in it
                mrirt.render_brats(pe, g, gl, ext=dict(synth.SHADE_EXT, layout="mod4", labelLayout="brick"))
            del gq
        del g, gl


def test_mod4_odd_dims_cameras_and_overlays_against_the_oracle(env):
    """The MOD4 grid on volumes whose sides are odd (the 2 x 2 x 2 bricks are padded), from cameras outside and inside the box, with
    weights that are not 1, both overlays as plain label grids and as label cells, and exact empty-space skipping: whole frames
    and live-sample counts against the oracle."""
    torch, mrirt, synth, oc = env["torch"], env["mrirt"], env["synth"], env["oc"]
    rng = np.random.default_rng(5)
    for dims, hw in (((37, 29, 23), (57, 75)), ((19, 44, 31), (64, 40))):
        vols = [synth.synth_volume(0, 77 + m, phase=0.2 * m, dims=dims) for m in range(4)]
        lab = synth.synth_labels(0, dims=dims)
        prd = np.roll(lab, 11).copy()
        g4 = mrirt.upload_mod4(vols, dims)
        plain = (mrirt.upload_grid(lab, dims, "brick"), mrirt.upload_grid(prd, dims, "brick"))
        cells = mrirt.upload_label_cells(lab, prd, dims)
        for case in range(4):
            p = synth.brats_scene(0, 0, 90, dims=dims, image_hw=hw, channels=4, show_seg=True, show_pred=bool(case & 1), intensity_alpha=3.0)
            p["volWeight"] = tuple(np.float32(v) for v in (1.0, 0.5, 2.0, 0.25))
            p["volEnabled"] = ((1, 1, 1, 1), (1, 0, 1, 1), (0, 1, 0, 0), (1, 1, 0, 1))[case]
            cam = mrirt.OrbitalCamera(initial_radius=(2.4, 0.35, 1.6, 3.0)[case], initial_phi=float(rng.uniform(0, 6.28)),
                                      initial_theta=float(rng.uniform(-1.2, 1.2)))
            p["eye"], p["U"], p["V"], p["W"] = cam.get_basis()
            ref, aux = oc.brats_main(p, vols, lab, prd if case & 1 else None, None, return_aux=True)
            a, sa = mrirt.render_brats(p, [g4] * 4, plain[0], plain[1] if case & 1 else None, stats=True)
            assert np.array_equal(a.cpu().numpy(), ref), (dims, case)
            assert sa["live_samples"] == aux["live_samples"]
            b = mrirt.render_brats(p, [g4], cells)                     # (bound once: the grid carries all four)
            assert torch.equal(a, b), (dims, case, "label cells")
            c, sc = mrirt.render_brats(p, [g4] * 4, plain[0], plain[1] if case & 1 else None, stats=True, skip=True)
            assert torch.equal(a, c) and sc == sa, (dims, case, "skip")


def test_frames_in_flight_on_slot_streams(env):
    """The N > 1 frame loop of bench.py: DEPTH frames in flight, slot s of FrameExchange on its own HIP stream (wait for the slot's
    previous exchange -> march into the slot's compact buffer -> submit).  Single process (the exchange is a device copy), three
    cameras cycling through three slots for nine frames: every de-tiled frame must be the whole-frame render of ITS camera — a
    slot reused too early, or a march ordered on the wrong stream, shows up as another camera's pixels."""
    torch, mrirt, synth = env["torch"], env["mrirt"], env["synth"]
    from mrirt import tiles
    n, image, tile, depth = 96, 320, 64, 3
    vol = synth.synth_volume(n)
    g = mrirt.upload_grid(vol, (n, n, n), "vga")
    ext = dict(synth.SHADE_EXT, layout="vga")
    cams = []
    for k in range(3):
        p = synth.brats_scene(n, image, 160, channels=1, intensity_alpha=8.0)
        cam = mrirt.OrbitalCamera(initial_radius=3.0, initial_phi=np.radians(60 + 25 * k), initial_theta=np.radians(20 - 30 * k))
        p["eye"], p["U"], p["V"], p["W"] = cam.get_basis()
        cams.append((p, mrirt.render_brats(p, [g], ext=ext)))
    skew = tiles.balanced_skew(image, tile, 1)
    ex = tiles.FrameExchange(image, image, tile, torch.float32, "cuda", depth=depth, dst=0, skew=skew)
    my_ext = tiles.shard_ext(ext, 0, 1, tile, skew)
    got = []
    torch.cuda.synchronize()
    for s in range(9):
        slot = s % depth
        with torch.cuda.stream(ex.stream(slot)):
            if s >= depth:
                got.append((s - depth, ex.finish(slot).clone()))
            mrirt.render_brats(cams[s % 3][0], [g], out=ex.local(slot), ext=my_ext)
            ex.submit(slot)
    for s in range(9 - depth, 9):
        with torch.cuda.stream(ex.stream(s % depth)):
            got.append((s, ex.finish(s % depth).clone()))
    torch.cuda.synchronize()
    assert len(got) == 9
    for s, frame in got:
        assert torch.equal(frame, cams[s % 3][1]), s
