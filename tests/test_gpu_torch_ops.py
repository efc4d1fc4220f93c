"""torch.ops.mrirt.* (mrirt/torch_ops.py): the registered operators launch the same C-ABI entry points
as the Python API, so their frames must be bit-identical to it — and through it to the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_render_brats_op_equals_api_and_oracle():
    import torch
    import mrirt
    from mrirt import synth, torch_ops
    from oracle import oracle_c
    n, image = 40, 96
    vol = synth.synth_volume(n)
    lab = synth.synth_labels(n)
    p = synth.brats_scene(n, image, 64, channels=1, intensity_alpha=8.0)
    p["showSeg"] = 1
    ref = oracle_c.brats_main(p, [vol], lab, None, dict(synth.SHADE_EXT))
    for layout in ("linear", "vg", "vga"):
        ext = dict(synth.SHADE_EXT, layout=layout, labelLayout="linear")
        g = mrirt.upload_grid(vol, (n, n, n), layout)
        labels = torch.from_numpy(lab.astype(np.int32)).cuda()
        api = mrirt.render_brats(p, [g], labels=labels, ext=ext)
        op = torch.ops.mrirt.render_brats(torch_ops.pack_brats_params(p), torch_ops.pack_render_ext(ext),
                                          g.data, None, None, None, labels, None)
        assert torch.equal(op, api)
        assert np.array_equal(op.cpu().numpy(), ref)
    # QUAD voxels + label cells (both overlays in one grid, bound as gLabels; gPreds unused), both operator libraries
    q = dict(p, showPred=1)
    pred = np.roll(lab, 7)
    refq = oracle_c.brats_main(q, [vol], lab, pred, None)
    gq = mrirt.upload_grid(vol, (n, n, n), "quad")
    cells = mrirt.upload_label_cells(lab, pred, (n, n, n))
    extq = dict(layout="quad", labelLayout="labcell")
    opq = torch.ops.mrirt.render_brats(torch_ops.pack_brats_params(q), torch_ops.pack_render_ext(extq), gq.data, None, None, None, cells.data, None)
    assert np.array_equal(opq.cpu().numpy(), refq)
    assert torch.equal(torch_ops.load_native().render_brats(torch_ops.pack_brats_params(q), torch_ops.pack_render_ext(extq),
                                                            gq.data, None, None, None, cells.data, None), opq)      # the C++ extension
    with pytest.raises(ValueError):
        torch.ops.mrirt.render_brats(torch_ops.pack_brats_params(q), torch_ops.pack_render_ext(extq), gq.data, None, None, None, cells.data[:64], None)
    # four modalities as ONE float4 grid (MOD4), bound as gIntensity0 alone; label cells; both operator libraries
    vols4 = [synth.synth_volume(n, 1234 + m, phase=0.3 * m) for m in range(4)]
    q4 = synth.brats_scene(n, image, 64, channels=4, show_seg=True, show_pred=True, intensity_alpha=4.0)
    ref4 = oracle_c.brats_main(q4, vols4, lab, pred, None)
    g4 = mrirt.upload_mod4(vols4, (n, n, n))
    ext4 = dict(layout="mod4", labelLayout="labcell")
    op4 = torch.ops.mrirt.render_brats(torch_ops.pack_brats_params(q4), torch_ops.pack_render_ext(ext4), g4.data, None, None, None, cells.data, None)
    assert np.array_equal(op4.cpu().numpy(), ref4)
    assert torch.equal(torch_ops.load_native().render_brats(torch_ops.pack_brats_params(q4), torch_ops.pack_render_ext(ext4),
                                                            g4.data, None, None, None, cells.data, None), op4)
    with pytest.raises(ValueError):
        torch.ops.mrirt.render_brats(torch_ops.pack_brats_params(q4), torch_ops.pack_render_ext(ext4), g4.data[:1000], None, None, None, cells.data, None)
    # the size checks the C ABI cannot make
    with pytest.raises(ValueError):
        torch.ops.mrirt.render_brats(torch_ops.pack_brats_params(p), torch_ops.pack_render_ext(ext),
                                     g.data[:100], None, None, None, labels, None)
    with pytest.raises(TypeError):
        torch.ops.mrirt.render_brats(torch_ops.pack_brats_params(p)[:-1], torch_ops.pack_render_ext(ext),
                                     g.data, None, None, None, labels, None)


def test_render_volume_and_sdf_ops():
    import torch
    import mrirt
    from mrirt import synth, torch_ops
    p = synth.volume_scene(32, 80, 48)
    vol_u8 = synth.synth_u8_volume(32)
    t = torch.from_numpy(np.ascontiguousarray(vol_u8.reshape(-1))).cuda()
    api = mrirt.render_volume_u8(p, t, mode="u8")
    op = torch.ops.mrirt.render_volume(torch_ops.pack_volume_params(p), torch_ops.pack_render_ext(None), t, 1)
    assert torch.equal(op, api)
    assert float(api[..., 0].max()) > 0.05
    # the reference's own upload format: one uint32 per voxel (app.py:149-153)
    tp = torch.from_numpy(mrirt.volume.pack_u8_as_u32x4(vol_u8).astype(np.int64).astype(np.int32)).cuda().reshape(-1)
    assert torch.equal(torch.ops.mrirt.render_volume(torch_ops.pack_volume_params(p), torch_ops.pack_render_ext(None), tp, 0), api)
    sp, eye, U, V, W = synth.sdf_scene()
    api3 = mrirt.render_sdf(sp, eye, U, V, W, 72, 40)
    op3 = torch.ops.mrirt.render_sdf(torch_ops.pack_sdf_params(sp, eye, U, V, W), 72, 40, api3)
    assert torch.equal(op3, api3)


def test_inr_forward_op_equals_api():
    import torch
    from mrirt import inr
    rng = np.random.default_rng(5)
    dims = [7, 64, 64, 4]
    params = [{"W": (rng.standard_normal((dims[i], dims[i + 1])) * 0.3).astype(np.float32),
               "b": (rng.standard_normal(dims[i + 1]) * 0.1).astype(np.float32)} for i in range(3)]
    net = inr.pack_mlp(params, inr.KIND_SIREN, 0, 4, w0=30.0)
    n = 1000
    coords = (torch.rand((n, 3), device="cuda") * 2 - 1).contiguous()
    feats = torch.rand((n, 4), device="cuda").contiguous()
    api, _ = inr._forward(net, coords, feats, n, True, False)
    op = torch.ops.mrirt.inr_forward(net.weights, net.biases, inr.KIND_SIREN, 3, 7, 4, 64, 0, 4, 30.0, coords, feats, n)
    assert torch.equal(op, api)
    from mrirt import torch_ops
    native = torch_ops.load_native().inr_forward(net.weights, net.biases, inr.KIND_SIREN, 3, 7, 4, 64, 0, 4, 30.0, coords, feats, n)
    assert torch.equal(native, api)
    with pytest.raises(ValueError):
        torch_ops.load_native().inr_forward(net.weights, net.biases, inr.KIND_SIREN, 3, 7, 4, 64, 0, 4, 30.0, coords[:10], feats, n)


def test_native_cpp_operators_match_the_python_registered_ones():
    """torch.ops.mrirt_native.* (csrc/torch_binding.cpp, the C++ extension) against torch.ops.mrirt.*: same frames."""
    import torch
    import mrirt
    from mrirt import synth, torch_ops
    ops = torch_ops.load_native()
    n, image = 40, 96
    vol, lab = synth.synth_volume(n), synth.synth_labels(n)
    p = synth.brats_scene(n, image, 64, channels=1, intensity_alpha=8.0)
    p["showSeg"] = 1
    labels = torch.from_numpy(lab.astype(np.int32)).cuda()
    for layout in ("linear", "vga", "quad"):
        ext = dict(synth.SHADE_EXT if layout == "vga" else {}, layout=layout, labelLayout="linear")
        g = mrirt.upload_grid(vol, (n, n, n), layout)
        args = (torch_ops.pack_brats_params(p), torch_ops.pack_render_ext(ext), g.data, None, None, None, labels, None)
        assert torch.equal(ops.render_brats(*args), torch.ops.mrirt.render_brats(*args))
    half = torch_ops.pack_render_ext(dict(layout="quad", labelLayout="linear", outFormat="rgba16f", tileSize=32, tileRank=1, tileWorld=2))
    a = ops.render_brats(torch_ops.pack_brats_params(p), half, g.data, None, None, None, labels, None)
    b = torch.ops.mrirt.render_brats(torch_ops.pack_brats_params(p), half, g.data, None, None, None, labels, None)
    assert a.dtype == torch.float16 and a.shape == b.shape and torch.equal(a, b)
    with pytest.raises(ValueError):
        ops.render_brats(torch_ops.pack_brats_params(p), torch_ops.pack_render_ext(ext), g.data[:100], None, None, None, labels, None)
    pv = synth.volume_scene(32, 80, 48)
    u8 = torch.from_numpy(synth.synth_u8_volume(32)).cuda()
    e0 = torch_ops.pack_render_ext({})
    assert torch.equal(ops.render_volume(torch_ops.pack_volume_params(pv), e0, u8, 1),
                       torch.ops.mrirt.render_volume(torch_ops.pack_volume_params(pv), e0, u8, 1))
    sp, eye, U, V, W = synth.sdf_scene()
    blob = torch_ops.pack_sdf_params(sp, eye, U, V, W)
    assert torch.equal(ops.render_sdf(blob, 96, 72, u8), torch.ops.mrirt.render_sdf(blob, 96, 72, u8))
