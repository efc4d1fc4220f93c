"""Exact empty-space skipping (mrirt_render_brats_skip): the frame and the counters must be the SAME BITS as
the plain launch — and as the oracle — on volumes that really have empty space, for every kernel family that
implements it, and on configurations where it must switch itself off."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def head_in_air(n, seed=3, channels=1):
    """A blob of texture inside zeros (like a skull-stripped scan), plus a label grid with one label blob
    INSIDE the tissue and one floating in the air (so labels alone must keep those macro cells alive)."""
    rng = np.random.default_rng(seed)
    ax = np.linspace(-1, 1, n, dtype=np.float32)
    z, y, x = np.meshgrid(ax, ax, ax, indexing="ij")
    r = np.sqrt(x * x + y * y + z * z)
    vols = []
    for c in range(channels):
        v = np.clip(0.9 - r, 0, None) * (0.8 + 0.2 * np.sin((5 + c) * x) * np.cos(4 * y)) + 0.03 * rng.random((n, n, n), dtype=np.float32)
        v[r > 0.62 + 0.03 * c] = 0.0
        vols.append(np.ascontiguousarray(v.astype(np.float32)).reshape(-1))
    lab = np.zeros((n, n, n), np.uint32)
    lab[r < 0.2] = 3
    lab[(np.abs(x - 0.8) < 0.08) & (np.abs(y + 0.75) < 0.08) & (np.abs(z - 0.7) < 0.08)] = 2      # in the air
    return vols, lab.reshape(-1)


def mask_fraction(dims):
    import ctypes as C
    import torch
    from mrirt import _lib, render
    d = (C.c_uint32 * 3)(*dims)
    cells = int(_lib.lib().mrirt_macro_cells(d))
    words = int(_lib.lib().mrirt_skip_mask_words(d))
    assert render._last_skip_mask.numel() == words
    m = render._last_skip_mask.cpu().numpy().view(np.uint32)
    bits = np.unpackbits(m.view(np.uint8), bitorder="little")[:cells]
    return float(bits.mean())


@pytest.mark.parametrize("layout,shade,channels", [("vg", True, 1), ("vg", False, 1), ("quad", False, 1), ("quad", False, 3), ("mod4", False, 3)])
@pytest.mark.parametrize("math", ["strict", "fast"])
def test_skip_is_bit_identical(layout, shade, channels, math):
    import torch
    import mrirt
    from mrirt import synth
    n, image = 72, 160
    vols, lab = head_in_air(n, channels=channels)
    p = synth.brats_scene(n, image, 192, channels=channels, intensity_alpha=6.0)
    p["wl"], p["ww"] = np.float32(0.45), np.float32(0.7)            # window floor 0.1 > the air's 0
    ext = dict(synth.SHADE_EXT) if shade else {}
    ext.update(layout=layout, math=math)
    if layout == "mod4":        # the (three enabled) modalities as one float4 grid; the fourth is one the frame never enables
        grids = [mrirt.upload_mod4(list(vols) + [None] * (4 - len(vols)), (n, n, n))] * 4
    else:
        grids = [mrirt.upload_grid(v, (n, n, n), layout) for v in vols]
    labels = mrirt.upload_grid(lab, (n, n, n), "linear")
    for show_seg in (0, 1):
        p["showSeg"] = np.uint32(show_seg)
        plain, st0 = mrirt.render_brats(p, grids, labels=labels if show_seg else None, ext=ext, stats=True)
        fast, st1 = mrirt.render_brats(p, grids, labels=labels if show_seg else None, ext=ext, stats=True, skip=True)
        assert torch.equal(plain, fast)
        assert st0 == st1
        frac = mask_fraction((n, n, n))
        assert frac > 0.5, f"only {frac:.2f} of the macro cells were skippable: the test volume is not doing its job"
        if show_seg:        # the label blob in the air must be visible, i.e. its cells were not skipped
            off = mrirt.render_brats({**p, "showSeg": np.uint32(0)}, grids, ext=ext)
            assert not torch.equal(off, plain)


def test_skip_matches_oracle_and_switches_itself_off():
    import torch
    import mrirt
    from mrirt import synth
    from oracle import oracle_c
    n, image = 48, 96
    vols, lab = head_in_air(n, seed=9)
    p = synth.brats_scene(n, image, 128, channels=1, intensity_alpha=6.0)
    p["wl"], p["ww"], p["showSeg"] = np.float32(0.45), np.float32(0.7), np.uint32(1)
    g = mrirt.upload_grid(vols[0], (n, n, n), "quad")
    labels = mrirt.upload_grid(lab, (n, n, n), "linear")
    ref = oracle_c.brats_main(p, [vols[0]], lab, None)
    got = mrirt.render_brats(p, [g], labels=labels, skip=True)
    assert np.array_equal(got.cpu().numpy(), ref)
    # configurations where "bound <= window floor" proves nothing: the launch must ignore the mask
    for change in ({"gamma": np.float32(0.5)}, {"ww": np.float32(-0.7)}, {"volWeight": (np.float32(-1.0), 0, 0, 0)}):
        q = {**p, **change}
        a = mrirt.render_brats(q, [g], labels=labels)
        b = mrirt.render_brats(q, [g], labels=labels, skip=True)
        assert torch.equal(a, b), change
    # layouts without the pipelined kernel simply render as before
    gl = mrirt.upload_grid(vols[0], (n, n, n), "linear")
    assert torch.equal(mrirt.render_brats(p, [gl], labels=labels, skip=True), got)
    # raw arrays cannot be skipped: they carry no macro summary
    with pytest.raises(ValueError):
        mrirt.render_brats(p, [vols[0]], labels=labels, skip=True)


def test_skip_really_skips_and_the_map_is_cached_across_frames():
    """A guard that skipping actually skips, on counters rather than on a stopwatch: with
    kernelVariant bit 7 the skipping kernels add the samples they did NOT fetch (flagged cell, or leapt over) to
    stats[1]; on a 256^3 blob-in-air most samples must be of that kind.  And the empty-radius map depends on window /
    weights / overlays, not on the camera: the second frame of an orbit must reuse it (no pre-pass launches)."""
    import torch
    import mrirt
    from mrirt import synth, render
    n, image = 256, 512
    vols, _ = head_in_air(n, seed=1)
    p = synth.brats_scene(n, image, 384, channels=1, intensity_alpha=2.0)
    p["wl"], p["ww"] = np.float32(0.45), np.float32(0.7)
    ext = dict(synth.SHADE_EXT, layout="vg")
    g = mrirt.upload_grid(vols[0], (n, n, n), "vg")
    plain, st0 = mrirt.render_brats(p, [g], ext=ext, stats=True)
    builds = render.skip_map_builds
    skipped, st1 = mrirt.render_brats(p, [g], ext=ext, stats=True, skip=True)
    assert torch.equal(plain, skipped) and st0 == st1
    assert render.skip_map_builds == builds + 1
    _, st2 = mrirt.render_brats(p, [g], ext=dict(ext, kernelVariant=128), stats=True, skip=True)
    unfetched = st2["shaded_samples"] - st0["shaded_samples"]
    assert st2["live_samples"] == st0["live_samples"]
    print(f"\n256^3 blob in air, 512^2 x 384 steps: {unfetched} of {st0['live_samples']} samples not fetched "
          f"({unfetched / st0['live_samples']:.3f})")
    assert unfetched >= 0.5 * st0["live_samples"]
    assert render.skip_map_builds == builds + 1, "same window / weights / overlays: the map is reused"
    # another camera: still the same map; the frame is still the plain kernel's
    cam = mrirt.OrbitalCamera(initial_radius=3.0, initial_phi=np.radians(60), initial_theta=np.radians(-40))
    eye, U, V, W = cam.get_basis()
    q = dict(p, eye=eye, U=U, V=V, W=W)
    a = mrirt.render_brats(q, [g], ext=ext)
    b = mrirt.render_brats(q, [g], ext=ext, skip=True)
    assert torch.equal(a, b) and render.skip_map_builds == builds + 1
    # a different window is a different map
    q2 = dict(q, wl=np.float32(0.5))
    assert torch.equal(mrirt.render_brats(q2, [g], ext=ext), mrirt.render_brats(q2, [g], ext=ext, skip=True))
    assert render.skip_map_builds == builds + 2


@pytest.mark.parametrize("layout", ["vg", "vga"])
@pytest.mark.parametrize("channels", [2, 3, 4])
@pytest.mark.parametrize("math", ["strict", "fast"])
def test_skip_with_several_shaded_modalities_rolling_kernel(layout, channels, math):
    """VG / VGA with 2-4 shaded modalities and no overlay: the rolling kernel skips at packet granularity —
    same frame, same counters, and most samples of a blob-in-air really are not fetched (kernelVariant bit 7 counter).
    With an overlay shown there is no skipping kernel for this configuration: skip=True must then cost nothing (no
    pre-pass) and change nothing."""
    import torch
    import mrirt
    from mrirt import synth, render
    n, image = 64, 128
    vols, lab = head_in_air(n, channels=channels)
    p = synth.brats_scene(n, image, 160, channels=channels, intensity_alpha=6.0)
    p["wl"], p["ww"] = np.float32(0.45), np.float32(0.7)
    ext = dict(synth.SHADE_EXT, layout=layout, math=math)
    grids = [mrirt.upload_grid(v, (n, n, n), layout) for v in vols]
    plain, st0 = mrirt.render_brats(p, grids, ext=ext, stats=True)
    fast, st1 = mrirt.render_brats(p, grids, ext=ext, stats=True, skip=True)
    assert torch.equal(plain, fast) and st0 == st1
    _, st2 = mrirt.render_brats(p, grids, ext=dict(ext, kernelVariant=128), stats=True, skip=True)
    unfetched = st2["shaded_samples"] - st0["shaded_samples"]
    assert st2["live_samples"] == st0["live_samples"] and unfetched >= 0.3 * st0["live_samples"], (unfetched, st0)
    # one step at a time (kernelVariant bit 8: no leaps) must give the same frame too
    slow = mrirt.render_brats(p, grids, ext=dict(ext, kernelVariant=256), skip=True)
    assert torch.equal(slow, plain)
    # overlay on: no SKIP kernel -> no pre-pass (the scratch stays as allocated), same frame
    labels = mrirt.upload_grid(lab, (n, n, n), "brick")
    q = dict(p, showSeg=np.uint32(1))
    a = mrirt.render_brats(q, grids, labels=labels, ext=ext)
    b = mrirt.render_brats(q, grids, labels=labels, ext=ext, skip=True)
    assert torch.equal(a, b)
    render._SKIP_MAPS.clear()
    builds = render.skip_map_builds
    mrirt.render_brats(q, grids, labels=labels, ext=ext, skip=True)
    assert len(render._SKIP_MAPS) == 0 and render.skip_map_builds == builds, "no map, no scratch, no cache entry (ADVICE r3)"


def test_skip_switches_itself_off_on_a_volume_without_empty_space():
    """The synthetic bench volume has no air: the map of its first skipping frame says so (one read-back when the map is
    first reused), and later frames take the plain kernels — same bits, no skipping overhead."""
    import torch
    import mrirt
    from mrirt import synth, render
    n = 64
    vol = synth.synth_volume(n)
    g = mrirt.upload_grid(vol, (n, n, n), "quad")
    p = synth.brats_scene(n, 128, 96, channels=1, intensity_alpha=4.0)
    plain = mrirt.render_brats(p, [g])
    render._SKIP_MAPS.clear()
    frames = [mrirt.render_brats(p, [g], skip=True) for _ in range(3)]
    assert all(torch.equal(f, plain) for f in frames)
    (entry,) = render._SKIP_MAPS.values()
    assert entry.built and entry.empty_fraction is not None and entry.empty_fraction < render.SKIP_MIN_EMPTY_FRACTION


def test_skip_with_tile_sharding():
    """Tiles of three 'ranks' rendered with skipping reassemble to the plain whole frame."""
    import torch
    import mrirt
    from mrirt import synth, tiles
    n, image = 64, 192
    vols, _ = head_in_air(n, seed=4)
    p = synth.brats_scene(n, image, 160, channels=1, intensity_alpha=6.0)
    p["wl"], p["ww"] = np.float32(0.45), np.float32(0.7)
    ext = dict(synth.SHADE_EXT, layout="vg")
    g = mrirt.upload_grid(vols[0], (n, n, n), "vg")
    whole = mrirt.render_brats(p, [g], ext=ext)
    world, tile = 3, 64
    parts = [mrirt.render_brats(p, [g], ext=tiles.shard_ext(ext, r, world, tile), skip=True) for r in range(world)]
    max_local = tiles.local_tile_count(image, image, tile, 0, world)
    gathered = torch.zeros((world, max_local, tile, tile, 4), device="cuda")
    for r, t in enumerate(parts):
        gathered[r, :t.shape[0]] = t
    assert torch.equal(mrirt.detile(gathered, image, image, tile, world), whole)


@pytest.mark.parametrize("layout,shade", [("vga", True), ("vg", True), ("quad", False)])
@pytest.mark.parametrize("math", ["strict", "fast"])
def test_second_level_leaps_are_bit_identical(layout, shade, math):
    """Second level: the packet crosses wide empty regions in leaps sized by the empty-radius map (the march's own
    `t += dt` chain and live counter, no per-step work).  A small off-centre blob in a 200 x 168 x 136 box (sides that
    are not multiples of 8) leaves most of the box leapable; frames and counters must equal the plain launch and the
    step-by-step launch (kernelVariant bit 8), for cameras inside and outside the box and for a step longer than a
    macro cell; the map itself is checked against a brute-force Chebyshev transform of the mask."""
    import torch
    import mrirt
    from mrirt import render, synth
    dims = (200, 168, 136)
    z, y, x = np.meshgrid(*[np.arange(d, dtype=np.float32) for d in dims[::-1]], indexing="ij")
    r2 = ((x - 140) / 30) ** 2 + ((y - 60) / 24) ** 2 + ((z - 90) / 20) ** 2
    rng = np.random.default_rng(9)
    vol = np.where(r2 < 1, 0.3 + 0.6 * rng.random(r2.shape, dtype=np.float32), 0).astype(np.float32).reshape(-1)
    grid = mrirt.upload_grid(vol, dims, layout)
    for steps, dist in ((320, 2.4), (320, 0.35), (40, 2.4)):
        p = synth.brats_scene(max(dims), 192, steps, channels=1, intensity_alpha=5.0, dims=dims)
        p["eye"] = (np.asarray(p["eye"], np.float32) * np.float32(dist / 2.4)).astype(np.float32)
        p["wl"], p["ww"] = np.float32(0.45), np.float32(0.7)
        ext = dict(synth.SHADE_EXT) if shade else {}
        ext.update(layout=layout, math=math)
        plain, st0 = mrirt.render_brats(p, [grid], ext=ext, stats=True)
        l1, st1 = mrirt.render_brats(p, [grid], ext=dict(ext, kernelVariant=256), stats=True, skip=True)
        l2, st2 = mrirt.render_brats(p, [grid], ext=ext, stats=True, skip=True)
        assert torch.equal(plain, l1) and torch.equal(plain, l2)
        assert st0 == st1 == st2
        assert st0["live_samples"] > 10 * max(st0["shaded_samples"], 1) or not shade
    # the distance map follows the mask words: check it against a brute-force Chebyshev transform of the mask
    import ctypes as C
    from mrirt import _lib
    d = (C.c_uint32 * 3)(*dims)
    cells = int(_lib.lib().mrirt_macro_cells(d))
    macro_words = ((cells + 63) // 64) * 2
    raw = render._last_skip_mask.cpu().numpy().view(np.uint32)
    mx, my, mz = [(v + 7) // 8 for v in dims]
    empty = np.unpackbits(raw[:macro_words].view(np.uint8), bitorder="little")[:cells].reshape(mz, my, mx).astype(bool)
    got = raw[macro_words:].view(np.uint8)[:cells].reshape(mz, my, mx)
    want = np.zeros_like(got)
    pad = 31
    big = np.ones((mz + 2 * pad, my + 2 * pad, mx + 2 * pad), bool)          # outside the grid: no constraint
    big[pad:pad + mz, pad:pad + my, pad:pad + mx] = empty
    ok = np.ones_like(empty)
    for r in range(1, 32):
        # radius r: every cell within r - 1 is empty
        k = r - 1
        box = np.ones_like(empty)
        for dz in range(-k, k + 1):
            for dy in range(-k, k + 1):
                sl = big[pad + dz:pad + dz + mz, pad + dy:pad + dy + my]
                run = np.ones_like(empty)
                for dx in range(-k, k + 1):
                    run &= sl[:, :, pad + dx:pad + dx + mx]
                box &= run
        ok &= box
        want[ok] = r
        if not ok.any():
            break
    assert np.array_equal(got, want)
    assert (got >= 2).mean() > 0.3


@pytest.mark.parametrize("layout,shade", [("vga", True), ("quad", False)])
def test_skip_on_tile_shards_reassembles_to_the_plain_frame(layout, shade):
    """Skipping inside the multi-GPU tile path: every rank's compact tile buffer rendered with skip=True, de-tiled,
    is the plain whole frame bit for bit (image sides that are not multiples of the tile), counters add up."""
    import torch
    import mrirt
    from mrirt import synth, tiles
    n, w, h, tile, world = 96, 200, 150, 64, 3
    vols, _ = head_in_air(n, channels=1)
    p = synth.brats_scene(n, 0, 160, image_hw=(h, w), channels=1, intensity_alpha=6.0)
    p["wl"], p["ww"] = np.float32(0.45), np.float32(0.7)
    ext = dict(synth.SHADE_EXT) if shade else {}
    ext.update(layout=layout)
    g = mrirt.upload_grid(vols[0], (n, n, n), layout)
    whole, sw = mrirt.render_brats(p, [g], ext=ext, stats=True)
    max_local = tiles.local_tile_count(w, h, tile, 0, world)
    gathered = torch.zeros((world, max_local, tile, tile, 4), device="cuda")
    live = 0
    for r in range(world):
        part, st = mrirt.render_brats(p, [g], ext=tiles.shard_ext(ext, r, world, tile), stats=True, skip=True)
        gathered[r, :part.shape[0]] = part
        live += st["live_samples"]
    assert torch.equal(mrirt.detile(gathered, w, h, tile, world), whole)
    assert live == sw["live_samples"]
