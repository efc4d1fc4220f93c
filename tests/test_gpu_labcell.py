"""The cell-packed label layout (MRIRT_LAYOUT_LABCELL, render.upload_label_cells): both overlays' nearest labels as one
8-byte element per cell of the QUAD grid.  sampleLabel's rounded voxel (inr/viewer/brats_rt.slang:78-83) is always one of
the sample's cell corners, so frames and counters must be the SAME BITS as with the reference's two label grids — through the
pipelined kernel (one label gather), the generic kernel, the skipping kernels and the shim — including labels the shader
ignores (>= 8), samples on the volume's border and cameras inside it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import mrirt
    from mrirt import synth
    from oracle import oracle_c
    assert torch.cuda.is_available()
    return mrirt, synth, oracle_c


def _labels(rng, dims, hi):
    """Blobby label grids with values 0..hi (hi > 7: labels the shader must ignore), x fastest."""
    X, Y, Z = dims
    z, y, x = np.meshgrid(np.linspace(-1, 1, Z), np.linspace(-1, 1, Y), np.linspace(-1, 1, X), indexing="ij")
    f = np.sin(3.1 * x + rng.uniform(0, 3)) * np.cos(2.3 * y + rng.uniform(0, 3)) + np.sin(2.7 * z + rng.uniform(0, 3))
    lab = np.clip(np.floor((f + 2.0) / 4.0 * (hi + 1)), 0, hi).astype(np.uint32)
    lab[(x * x + y * y + z * z) > 1.4] = 0
    return lab.reshape(-1)


def test_label_cells_hold_the_corner_labels(env):
    mrirt, synth, oc = env
    rng = np.random.default_rng(5)
    dims = (11, 7, 6)
    seg, pred = _labels(rng, dims, 9), _labels(rng, dims, 5)
    g = mrirt.upload_label_cells(seg, pred, dims)
    assert g.layout == "labcell" and g.macro is not None and g.macro2 is not None
    words = g.data.cpu().numpy().view(np.uint32).reshape(-1, 2)
    X, Y, Z = dims
    nbx, nby = (X + 1) // 2, (Y + 1) // 2
    s3, p3 = seg.reshape(Z, Y, X), pred.reshape(Z, Y, X)
    for (x, y, z) in [(0, 0, 0), (X - 1, Y - 1, Z - 1), (3, 2, 1), (X - 2, 0, Z - 2), (4, Y - 1, 3)]:
        e = (((z >> 1) * nby + (y >> 1)) * nbx + (x >> 1)) * 8 + (x & 1) + 2 * (y & 1) + 4 * (z & 1)
        for c in range(8):
            xx, yy, zz = min(x + (c & 1), X - 1), min(y + ((c >> 1) & 1), Y - 1), min(z + (c >> 2), Z - 1)
            assert (int(words[e, 0]) >> (4 * c)) & 15 == min(int(s3[zz, yy, xx]), 8)
            assert (int(words[e, 1]) >> (4 * c)) & 15 == min(int(p3[zz, yy, xx]), 8)
    only = mrirt.upload_label_cells(None, pred, dims)
    assert not only.data.cpu().numpy().view(np.uint32).reshape(-1, 2)[:, 0].any() and only.macro is None


@pytest.mark.parametrize("seed", range(12))
def test_label_cells_render_the_same_bits(env, seed):
    mrirt, synth, oc = env
    import torch
    rng = np.random.default_rng(8800 + seed)
    dims = tuple(int(v) for v in rng.integers(14, 48, 3))
    nmod = 1 + seed % 4
    vols = [synth.synth_volume(0, 60 + 5 * seed + m, phase=0.4 * m, dims=dims) for m in range(4)]
    seg, pred = _labels(rng, dims, 9), _labels(rng, dims, 7)
    cam = synth.bench_camera(radius=float(rng.choice([0.3, 1.1, 2.4, 3.2])), phi_deg=float(rng.uniform(8, 172)), theta_deg=float(rng.uniform(0, 360)))
    show_seg, show_pred = [(True, True), (True, False), (False, True)][seed % 3]
    p = synth.brats_scene(0, 0, int(rng.integers(30, 220)), dims=dims, image_hw=(int(rng.integers(9, 80)), int(rng.integers(9, 80))),
                          channels=nmod, show_seg=show_seg, show_pred=show_pred, intensity_alpha=float(rng.choice([0.4, 6.0])),
                          camera=cam, fov_deg=float(rng.uniform(25, 80)))
    p["volWeight"] = tuple(float(v) for v in rng.uniform(0.3, 1.4, 4))
    p["gamma"] = float(rng.choice([1.0, 1.0, 1.6]))
    ref, aux = oc.brats_main(p, vols, seg if show_seg else None, pred if show_pred else None, None, return_aux=True)
    grids = [mrirt.upload_grid(v, dims, "quad") for v in vols]
    cells = mrirt.upload_label_cells(seg if show_seg else None, pred if show_pred else None, dims)
    for variant, skip in ((0, False), (4, False), (0, True)):        # pipelined (one label gather), generic, skipping
        got, st = mrirt.render_brats(p, grids, labels=cells, ext=dict(layout="quad", kernelVariant=variant), stats=True, skip=skip)
        assert np.array_equal(got.cpu().numpy(), ref), (seed, variant, skip, dims, float(np.abs(got.cpu().numpy() - ref).max()))
        assert st["live_samples"] == aux["live_samples"]
    # fast math: the label path is integer — the same frame as with the two brick grids
    gl, gp = mrirt.upload_grid(seg, dims, "brick"), mrirt.upload_grid(pred, dims, "brick")
    a = mrirt.render_brats(p, grids, labels=cells, ext=dict(layout="quad", math="fast"))
    b = mrirt.render_brats(p, grids, labels=gl if show_seg else None, preds=gp if show_pred else None, ext=dict(layout="quad", math="fast"))
    assert torch.equal(a, b)


def test_label_cells_at_config_2_and_through_the_shim(env):
    """BASELINE config 2 (256^3 x 4 + seg, 512^2, 256 steps) with label cells == with the brick label grid (which the full-size
    test holds to the oracle); the shim picks the cell grid by itself for QUAD frames with overlays."""
    mrirt, synth, oc = env
    import torch
    import mrirt.shim as spy
    n = 256
    vols = [synth.synth_volume(n, 1234 + m, phase=0.3 * m) for m in range(4)]
    lab = synth.synth_labels(n)
    p = synth.brats_scene(n, 512, 256, channels=4, show_seg=True, intensity_alpha=0.4)
    grids = [mrirt.upload_grid(v, (n, n, n), "quad") for v in vols]
    a, sa = mrirt.render_brats(p, grids, labels=mrirt.upload_grid(lab, (n, n, n), "brick"), stats=True)
    b, sb = mrirt.render_brats(p, grids, labels=mrirt.upload_label_cells(lab, None, (n, n, n)), stats=True)
    assert torch.equal(a, b) and sa == sb
    # the shim: same dict, same buffers as the viewer binds them
    dims = (40, 36, 30)
    vs = [synth.synth_volume(0, 7 + m, dims=dims) for m in range(4)]
    rng = np.random.default_rng(3)
    seg, pred = _labels(rng, dims, 8), _labels(rng, dims, 6)
    q = synth.brats_scene(0, 0, 90, dims=dims, image_hw=(72, 96), channels=4, show_seg=True, show_pred=True, intensity_alpha=2.0)
    ref = oc.brats_main(q, vs, seg, pred, None)
    dev = spy.Device(layout="auto")
    k = dev.create_compute_kernel(dev.load_program("brats_rt.slang", ["brats_main"]))
    bufs = []
    for arr in vs + [seg, pred]:
        bb = dev.create_buffer(element_count=arr.size, struct_size=4)
        bb.copy_from_numpy(arr)
        bufs.append(bb)
    tex = dev.create_texture(format=spy.Format.rgba32_float, width=96, height=72)
    vars_ = {"gOutput": tex, "gParams": q, "gLabels": bufs[4], "gPreds": bufs[5], **{f"gIntensity{m}": bufs[m] for m in range(4)}}
    k.dispatch(thread_count=[96, 72, 1], vars=vars_)
    assert len(dev._label_cells) == 1                                 # the cell grid was used (and cached)
    assert np.array_equal(tex.tensor.cpu().numpy(), ref)
    assert len(dev._mod4) == 1                                        # ... and the four modalities as ONE float4 grid (MOD4)
    bufs[5].copy_from_numpy(np.roll(pred, 5))                         # a rewritten prediction buffer must not meet a stale cell grid
    k.dispatch(thread_count=[96, 72, 1], vars=vars_)
    assert np.array_equal(tex.tensor.cpu().numpy(), oc.brats_main(q, vs, seg, np.roll(pred, 5), None))
    # two of the four switched off, their slots bound to the viewer's one-element dummies (brats_viewer.py:247-248): the MOD4
    # grid is rebuilt from the two that remain; then a rewritten modality must not meet a stale grid either
    dummy = dev.create_buffer(element_count=1, struct_size=4)
    dummy.copy_from_numpy(np.zeros(1, np.float32))
    q2 = dict(q, volEnabled=(1, 0, 1, 0))
    vars2 = dict(vars_, gParams=q2, gIntensity1=dummy, gIntensity3=dummy)
    k.dispatch(thread_count=[96, 72, 1], vars=vars2)
    assert np.array_equal(tex.tensor.cpu().numpy(), oc.brats_main(q2, vs, seg, np.roll(pred, 5), None))
    bufs[2].copy_from_numpy(vs[2][::-1].copy())
    k.dispatch(thread_count=[96, 72, 1], vars=vars2)
    assert np.array_equal(tex.tensor.cpu().numpy(), oc.brats_main(q2, [vs[0], vs[1], vs[2][::-1].copy(), vs[3]], seg, np.roll(pred, 5), None))


def test_label_cells_refuse_what_they_cannot_serve(env):
    mrirt, synth, oc = env
    dims = (20, 18, 16)
    vol = synth.synth_volume(0, 3, dims=dims)
    lab = synth.synth_labels(0, dims=dims)
    cells = mrirt.upload_label_cells(lab, lab, dims)
    p = synth.brats_scene(0, 0, 40, dims=dims, image_hw=(16, 16), channels=1, show_seg=True)
    with pytest.raises(ValueError):
        mrirt.render_brats(p, [mrirt.upload_grid(vol, dims, "vg")], labels=cells, ext=dict(layout="vg"))
    with pytest.raises(ValueError):
        mrirt.upload_label_cells(lab[:-1], None, dims)


def test_label_cells_with_real_empty_space_skipping(env):
    """A head in air (most macro cells skippable, one label blob floating in the air): the skipping kernels read the cells
    through fetch_labels — same bits and counters as the plain launch and as the two-grid path."""
    mrirt, synth, oc = env
    import torch
    from test_gpu_skip import head_in_air, mask_fraction
    n, image = 72, 160
    vols, lab = head_in_air(n, channels=2)
    pred = np.roll(lab, 11).copy()
    p = synth.brats_scene(n, image, 192, channels=2, show_seg=True, show_pred=True, intensity_alpha=6.0)
    p["wl"], p["ww"] = np.float32(0.45), np.float32(0.7)
    grids = [mrirt.upload_grid(v, (n, n, n), "quad") for v in vols]
    cells = mrirt.upload_label_cells(lab, pred, (n, n, n))
    gl, gp = mrirt.upload_grid(lab, (n, n, n), "linear"), mrirt.upload_grid(pred, (n, n, n), "linear")
    two, s2 = mrirt.render_brats(p, grids, labels=gl, preds=gp, ext=dict(layout="quad"), stats=True)
    for math in ("strict", "fast"):
        ext = dict(layout="quad", math=math)
        plain, s0 = mrirt.render_brats(p, grids, labels=cells, ext=ext, stats=True)
        skipped, s1 = mrirt.render_brats(p, grids, labels=cells, ext=ext, stats=True, skip=True)
        assert torch.equal(plain, skipped) and s0 == s1
        assert mask_fraction((n, n, n)) > 0.5
        if math == "strict":
            assert torch.equal(plain, two) and s0 == s2
