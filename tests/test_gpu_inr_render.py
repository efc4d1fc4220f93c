"""BASELINE config 5 (build-defined): K1 with the prediction label queried from the MLP at every
sample.  Checked in three layers so the bf16 classifier does not blur the exact parts:
  1. sample counts and emitted MLP inputs == the oracle's, exactly (fp32 / fp64 coordinate);
  2. compositing with the GPU's own class stream == the oracle compositing with that stream;
  3. end to end vs the all-fp32 oracle: classes agree on >= 99.9 % of the samples (near logit ties of the bf16 pass are
     re-evaluated in split bf16), >= 99.5 % of the pixels are within 1e-4, the rest within one overlay step."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    import torch
    import mrirt
    from mrirt import synth
    from oracle import oracle_np as onp
    assert torch.cuda.is_available()
    dims = (24, 20, 18)
    vols = [synth.synth_volume(0, 1234 + m, phase=0.3 * m, dims=dims) for m in range(4)]
    lab = synth.synth_labels(0, dims=dims)
    rng = np.random.default_rng(5)
    K, hidden = 4, 64
    sizes = [3 + 6 * K + 4] + [hidden] * 4 + [4]
    mlp = [{"W": (rng.uniform(-1, 1, (sizes[i], sizes[i + 1])) * np.sqrt(6 / (sizes[i] + sizes[i + 1]))).astype(np.float32),
            "b": rng.uniform(-0.2, 0.2, sizes[i + 1]).astype(np.float32)} for i in range(5)]
    zmu = [float(v[v != 0].mean()) for v in vols]
    zsg = [float(v[v != 0].std() + 1e-6) for v in vols]
    p = synth.brats_scene(0, 0, 48, dims=dims, image_hw=(40, 56), channels=2, show_seg=True, show_pred=True,
                          intensity_alpha=6.0)
    return dict(torch=torch, mrirt=mrirt, onp=onp, dims=dims, vols=vols, lab=lab, mlp=mlp, K=K, zmu=zmu, zsg=zsg, p=p)


@pytest.mark.parametrize("layout", ["linear", "quad"])
def test_c5_layers(setup, layout):
    s = setup
    mrirt, onp, torch = s["mrirt"], s["onp"], s["torch"]
    dims, p = s["dims"], s["p"]
    net = mrirt.inr.pack_mlp(s["mlp"], mrirt.inr.KIND_FOURIER_RELU, s["K"], 4)
    grids = [mrirt.upload_grid(v, dims, layout) for v in s["vols"]]
    gl = mrirt.upload_grid(s["lab"], dims, "linear" if layout == "linear" else "brick")
    img, aux = mrirt.inr.render_brats_inr(p, grids, net, s["zmu"], s["zsg"], labels=gl, return_aux=True, one_pass=True)
    img = img.cpu().numpy()

    # -- 1. counts and inputs ------------------------------------------------------------------------
    _, a0 = onp.brats_main(dict(p, showPred=0), s["vols"], s["lab"], None, dict(ertThreshold=-1.0), return_aux=True)
    counts = aux["counts"].cpu().numpy().reshape(a0["nsteps"].shape)
    assert np.array_equal(counts, a0["nsteps"]), "steps per ray with ERT off"
    assert aux["queries"] == int(a0["nsteps"].sum()) and aux["queries"] > 10000

    # oracle inputs: replay every sample position through the oracle's helper
    offsets = aux["offsets"].cpu().numpy()
    rec_c, rec_f = np.zeros((aux["queries"], 3), np.float32), np.zeros((aux["queries"], 4), np.float32)

    def recorder(idx, k, qx, qy, qz):
        c, f = onp.inr_sample_inputs(s["vols"], dims, s["zmu"], s["zsg"], qx, qy, qz)
        rows = offsets[idx] + k
        rec_c[rows], rec_f[rows] = c, f
        return np.zeros(idx.size, np.int64)

    onp.brats_main(dict(p, showSeg=0), s["vols"], None, recorder, dict(ertThreshold=-1.0))
    assert np.array_equal(aux["coords"].cpu().numpy(), rec_c), "sample coordinates"
    assert np.array_equal(aux["feats"].cpu().numpy(), rec_f), "z-scored trilinear intensities"

    # -- 2. compositing with the GPU's class stream is exact ------------------------------------------
    classes = aux["classes"].cpu().numpy()
    ref_stream = onp.brats_main_inr(p, s["vols"], None, 0, s["zmu"], s["zsg"], labels=s["lab"],
                                    class_stream=classes, ray_offsets=offsets)
    assert np.abs(img - ref_stream).max() <= 1e-6

    # -- 3. end to end vs the fp32 oracle ---------------------------------------------------------------
    x = onp.build_input(rec_c, rec_f, s["K"])
    logits = onp.apply_mlp(s["mlp"], x)
    want = np.argmax(logits, axis=-1)
    agree = (classes == want)
    assert agree.mean() >= 0.999, agree.mean()                       # SURVEY 8c; near-ties are re-evaluated in split bf16
    top2 = np.sort(logits[~agree], axis=-1)[:, -2:]
    assert np.all(top2[:, 1] - top2[:, 0] <= 1e-4 * np.abs(logits).max()), "a remaining disagreement is a tie at the fp32 oracle's own noise level"
    ref = onp.brats_main_inr(p, s["vols"], s["mlp"], s["K"], s["zmu"], s["zsg"], labels=s["lab"])
    d = np.abs(img - ref)[..., :3].max(axis=-1)
    # a ray is exact unless one of its ~40 samples flipped
    flipped_rays = sum(1 for o, c in zip(offsets, counts.reshape(-1)) if c and not agree[o:o + c].all()) / counts.size
    assert (d > 1e-4).mean() <= flipped_rays + 1e-9, "only rays holding a flipped class may differ"
    assert (d <= 1e-4).mean() >= 0.995, (d <= 1e-4).mean()          # BASELINE's 1e-4 on >= 99.5 % of the pixels
    assert d.max() <= 0.25          # a flipped overlay step: alpha*T*|lut.rgb| with alpha = 1 - e^{-0.9*dt*1.5}
    # the same frame with the refinement switched off shows what it buys (and that the switch works)
    raw, araw = mrirt.inr.render_brats_inr(p, grids, mrirt.inr.with_flags(net, no_refine=True), s["zmu"], s["zsg"], labels=gl,
                                           return_aux=True, one_pass=True)
    agree_raw = araw["classes"].cpu().numpy() == want
    assert agree_raw.mean() < agree.mean() and agree_raw.mean() >= 0.99


def test_c5_lattice_coordinates_are_predict_volumes(setup):
    """At lattice points the emitted coordinate is exactly predict_volume's (model.py:124-128)."""
    onp = setup["onp"]
    dims = (7, 5, 4)
    vols = [np.zeros(140, np.float32)] * 4
    for (i, j, k) in ((0, 0, 0), (6, 4, 3), (3, 2, 1), (5, 0, 2)):
        c, _ = onp.inr_sample_inputs(vols, dims, [0] * 4, [1] * 4, np.float32([i]), np.float32([j]), np.float32([k]))
        want = ((np.array([i, j, k]) / np.array([6, 4, 3])) * 2.0 - 1.0).astype(np.float32)
        assert np.array_equal(c[0], want)


def test_c5_argument_errors(setup):
    s = setup
    mrirt = s["mrirt"]
    net = mrirt.inr.pack_mlp(s["mlp"], mrirt.inr.KIND_FOURIER_RELU, s["K"], 4)
    with pytest.raises(ValueError):
        mrirt.inr.render_brats_inr(dict(s["p"], showPred=0), s["vols"], net, s["zmu"], s["zsg"], labels=s["lab"])
    with pytest.raises(ValueError):
        mrirt.inr.render_brats_inr(s["p"], s["vols"][:3], net, s["zmu"], s["zsg"], labels=s["lab"])


def _siren_params(rng, dims, w0=30.0):
    """siren_init of neumors_inr.ipynb:1150-1163: U(-r, r), r = sqrt(6 / fan_in) / (w0 on the first layer);
    small random biases instead of zeros so that the bias path is exercised."""
    out = {}
    for i in range(len(dims) - 1):
        r = np.sqrt(6.0 / dims[i]) / (w0 if i == 0 else 1.0)
        out[f"l{i}"] = {"w": rng.uniform(-r, r, (dims[i], dims[i + 1])).astype(np.float32),
                        "b": rng.uniform(-0.05, 0.05, dims[i + 1]).astype(np.float32)}
    return out


def _as_list(siren):
    return [{"W": siren[f"l{i}"]["w"], "b": siren[f"l{i}"]["b"]} for i in range(len(siren))]


@pytest.mark.parametrize("kind", ["fourier", "siren"])
def test_c5_chunked_is_bit_identical_to_one_pass(setup, kind):
    """mrirt_render_brats_inr (ERT-aware passes of chunk_steps) against the whole-ray three-pass form: the same
    frame bit for bit at every chunk size, fewer MLP queries once rays terminate early, and the live-sample
    counter equal to the composited samples of the one-pass march."""
    s = setup
    mrirt, torch = s["mrirt"], s["torch"]
    inr = mrirt.inr
    dims = s["dims"]
    rng = np.random.default_rng(41)
    if kind == "fourier":
        net = inr.pack_mlp(s["mlp"], inr.KIND_FOURIER_RELU, s["K"], 4)
    else:
        net = inr.pack_mlp(_as_list(_siren_params(rng, [7, 64, 64, 4])), inr.KIND_SIREN, 0, 4)
    # dense preset so that early termination really fires, 3 enabled modalities + both overlays
    p = dict(s["p"], intensityAlpha=40.0, volEnabled=(1, 1, 1, 0))
    grids = [mrirt.upload_grid(v, dims, "quad") for v in s["vols"]]
    gl = mrirt.upload_grid(s["lab"], dims, "brick")
    ref, a0 = inr.render_brats_inr(p, grids, net, s["zmu"], s["zsg"], labels=gl, return_aux=True, one_pass=True)
    assert a0["live_samples"] < a0["queries"], "the scene must terminate some rays early"
    seen = set()
    for chunk in (1, 5, 32, 4096):
        img, a = inr.render_brats_inr(p, grids, net, s["zmu"], s["zsg"], labels=gl, return_aux=True, chunk_steps=chunk)
        assert torch.equal(img, ref), chunk
        assert a["live_samples"] == a0["live_samples"]
        assert a0["live_samples"] <= a["queries"] <= a0["queries"]
        seen.add(a["queries"])
    # the default pass length follows the scene: 32 where intensity alone can terminate a ray (this dense preset), 96 where it
    # cannot (the reference viewer's slider range) — same bits either way
    img, a = inr.render_brats_inr(p, grids, net, s["zmu"], s["zsg"], labels=gl, return_aux=True)
    assert a["chunk_steps"] == 32 and torch.equal(img, ref)
    thin = dict(p, intensityAlpha=0.4)
    ref_thin = inr.render_brats_inr(thin, grids, net, s["zmu"], s["zsg"], labels=gl, one_pass=True)
    img, a = inr.render_brats_inr(thin, grids, net, s["zmu"], s["zsg"], labels=gl, return_aux=True)
    assert a["chunk_steps"] == 96 and torch.equal(img, ref_thin)
    img1, a1 = inr.render_brats_inr(p, grids, net, s["zmu"], s["zsg"], labels=gl, return_aux=True, chunk_steps=1)
    assert a1["queries"] == a0["live_samples"], "one step per pass classifies exactly the live samples"
    assert len(seen) > 1
    # strict and fast math both run; a linear-layout bind too
    lin = [mrirt.upload_grid(v, dims, "linear") for v in s["vols"]]
    img2 = inr.render_brats_inr(p, lin, net, s["zmu"], s["zsg"], labels=mrirt.upload_grid(s["lab"], dims, "linear"), chunk_steps=16)
    assert torch.equal(img2, ref)
    fast = inr.render_brats_inr(p, grids, net, s["zmu"], s["zsg"], labels=gl, ext=dict(math="fast"), chunk_steps=16)
    assert float((fast - ref).abs().max()) < 0.3 and float((fast - ref).abs().mean()) < 2e-3
    # gradient shading (VG voxels: the emission pass also records the sample's gradient) and bricked fp32 voxels
    from mrirt import synth
    vg = [mrirt.upload_grid(v, dims, "vg") for v in s["vols"]]
    ext = dict(synth.SHADE_EXT, layout="vg")
    ref_sh, ash = inr.render_brats_inr(p, vg, net, s["zmu"], s["zsg"], labels=gl, ext=ext, return_aux=True, one_pass=True)
    assert not torch.equal(ref_sh, ref)
    for chunk in (3, 32):
        img, a = inr.render_brats_inr(p, vg, net, s["zmu"], s["zsg"], labels=gl, ext=ext, return_aux=True, chunk_steps=chunk)
        assert torch.equal(img, ref_sh), chunk
        assert a["live_samples"] == ash["live_samples"] and a["shaded_samples"] == ash["shaded_samples"]
    br = [mrirt.upload_grid(v, dims, "brick") for v in s["vols"]]
    assert torch.equal(inr.render_brats_inr(p, br, net, s["zmu"], s["zsg"], labels=gl, chunk_steps=7), ref)
    # the four modalities as ONE float4 grid (MRIRT_LAYOUT_MOD4): same bits, same counters, at every pass length
    m4 = mrirt.upload_mod4(s["vols"], dims)
    for chunk in (5, 32, None):
        img, a = inr.render_brats_inr(p, m4, net, s["zmu"], s["zsg"], labels=gl, return_aux=True, chunk_steps=chunk)
        assert torch.equal(img, ref), chunk
        assert a["live_samples"] == a0["live_samples"]
    assert torch.equal(inr.render_brats_inr(dict(p, volEnabled=(0, 1, 0, 1)), m4, net, s["zmu"], s["zsg"], labels=gl),
                       inr.render_brats_inr(dict(p, volEnabled=(0, 1, 0, 1)), grids, net, s["zmu"], s["zsg"], labels=gl))
    with pytest.raises(ValueError):                      # the whole-ray form marches with the K1 kernels
        inr.render_brats_inr(p, m4, net, s["zmu"], s["zsg"], labels=gl, one_pass=True)
    with pytest.raises(RuntimeError):                    # no gradients in it: MRIRT_ERR_LAYOUT
        inr.render_brats_inr(p, m4, net, s["zmu"], s["zsg"], labels=gl, ext=dict(synth.SHADE_EXT, layout="mod4"))
    # (K1 itself on a MOD4 grid: tests/test_gpu_parity.py, test_c2_full_size_against_the_oracle)
    # an orthographic camera: every ray has its own ORIGIN (the emission then reads it from the per-ray record the plan kernel
    # wrote; under a perspective camera it is the eye) — chunked passes on both bindings against the whole-ray form
    eo = dict(cameraMode=1, orthoHalfHeight=0.8)
    ref_o = inr.render_brats_inr(p, grids, net, s["zmu"], s["zsg"], labels=gl, ext=eo, one_pass=True)
    assert not torch.equal(ref_o, ref)
    for g in (grids, m4):
        for chunk in (7, 32):
            assert torch.equal(inr.render_brats_inr(p, g, net, s["zmu"], s["zsg"], labels=gl, ext=eo, chunk_steps=chunk), ref_o), chunk
    # no seg overlay bound at all (the emission then records no seg labels)
    p3 = dict(p, showSeg=0)
    ref3 = inr.render_brats_inr(p3, grids, net, s["zmu"], s["zsg"], labels=None, one_pass=True)
    assert torch.equal(inr.render_brats_inr(p3, grids, net, s["zmu"], s["zsg"], labels=None, chunk_steps=9), ref3)
    assert not torch.equal(ref3, ref)
    # an image whose sides are not multiples of the 8 x 8 packet (lanes outside the image take part in the wave's
    # row numbering with zero samples)
    p2 = dict(p, imageSize=(37, 29))
    ref2 = inr.render_brats_inr(p2, grids, net, s["zmu"], s["zsg"], labels=gl, one_pass=True)
    assert ref2.shape[:2] == (29, 37)
    for chunk in (2, 32):
        assert torch.equal(inr.render_brats_inr(p2, grids, net, s["zmu"], s["zsg"], labels=gl, chunk_steps=chunk), ref2)


def test_c5_as_named_siren_4x256_512x512_256_samples():
    """BASELINE config 5 as stated: 4 x 256 SIREN (7 -> 256 x 4 -> 4, neumors_inr.ipynb:853-899,1165-1178) queried
    per sample, 512 x 512, 256 samples/ray, on the 256^3 four-modality scene.
      * the chunked ERT-aware frame == the whole-ray three-pass frame, bit for bit, with fewer queries;
      * on a band of rows: sample counts and emitted inputs == the oracle's exactly; compositing with the GPU's
        own class stream == the oracle's compositing of that stream exactly; the bf16 MFMA classes against the
        fp32 oracle: every disagreement is a near-tie of the top two fp32 logits."""
    import torch
    import mrirt
    from mrirt import synth, inr
    from oracle import oracle_np as onp
    n, image, steps = 256, 512, 256
    vols = [synth.synth_volume(n, 1234 + m, phase=0.3 * m) for m in range(4)]
    lab = synth.synth_labels(n)
    rng = np.random.default_rng(2)
    siren = _siren_params(rng, [7, 256, 256, 256, 256, 4])
    net = inr.pack_mlp(_as_list(siren), inr.KIND_SIREN, 0, 4)
    zmu = [float(v[v != 0].mean()) for v in vols]
    zsg = [float(v[v != 0].std() + 1e-6) for v in vols]
    p = synth.brats_scene(n, image, steps, channels=4, show_seg=True, show_pred=True, intensity_alpha=6.0)
    grids = [mrirt.upload_grid(v, (n, n, n), "quad") for v in vols]
    gl = mrirt.upload_grid(lab, (n, n, n), "brick")
    ref, a0 = inr.render_brats_inr(p, grids, net, zmu, zsg, labels=gl, return_aux=True, one_pass=True)
    img, a = inr.render_brats_inr(p, grids, net, zmu, zsg, labels=gl, return_aux=True)            # 32-step passes
    assert torch.equal(img, ref)
    assert a["live_samples"] == a0["live_samples"] and a["live_samples"] <= a["queries"] < a0["queries"]
    assert a0["queries"] > 15_000_000
    # ---- the oracle on a band of rows through the middle of the frame ------------------------------------
    r0, r1 = 252, 258
    W = image
    offsets = a0["offsets"].cpu().numpy()
    counts = a0["counts"].cpu().numpy()
    classes = a0["classes"].cpu().numpy()
    band_off = offsets[r0 * W:r1 * W]
    rec = {"c": np.zeros((int(counts[r0 * W:r1 * W].sum()), 3), np.float32), "f": None, "l": None}
    rec["f"] = np.zeros((rec["c"].shape[0], 4), np.float32)
    rec["l"] = np.zeros((rec["c"].shape[0], 4), np.float32)
    base = int(band_off[0])

    def record(idx, k, c, f, logits):
        rows = band_off[idx] + k - base
        rec["c"][rows], rec["f"][rows], rec["l"][rows] = c, f, logits

    got_band = ref[r0:r1].cpu().numpy()
    # compositing of the GPU's class stream (ERT on, as rendered); the recorder sees every composited sample
    ref_stream = onp.brats_main_inr(p, vols, siren, 0, zmu, zsg, labels=lab, kind="siren", rows=(r0, r1),
                                    class_stream=classes, ray_offsets=band_off, record=record)
    assert np.array_equal(got_band, ref_stream), np.abs(got_band - ref_stream).max()
    # inputs of every composited sample of the band, exactly
    _, aux_o = onp.brats_main(dict(p, showPred=0), vols, lab, None, dict(ertThreshold=-1.0), return_aux=True, rows=(r0, r1))
    assert np.array_equal(counts[r0 * W:r1 * W].reshape(r1 - r0, W), aux_o["nsteps"]), "steps per ray in [t0, t1)"
    lo, hi = base, base + rec["c"].shape[0]
    seen = np.abs(rec["l"]).sum(axis=1) != 0                  # rows the (ERT-limited) oracle march visited
    assert seen.mean() > 0.5
    assert np.array_equal(a0["coords"][lo:hi].cpu().numpy()[seen], rec["c"][seen])
    assert np.array_equal(a0["feats"][lo:hi].cpu().numpy()[seen], rec["f"][seen])
    # classes vs the fp32 oracle
    want = np.argmax(rec["l"][seen], axis=-1)
    have = classes[lo:hi][seen]
    agree = have == want
    top2 = np.sort(rec["l"][seen][~agree], axis=-1)[:, -2:]
    scale = np.abs(rec["l"][seen]).max()
    assert agree.mean() >= 0.999, agree.mean()                       # SURVEY 8c (0.9967 before the near-tie refinement)
    assert np.all(top2[:, 1] - top2[:, 0] <= 1e-4 * scale), "a remaining disagreement is a tie at the fp32 oracle's own noise level"
    # pixels of the band against the all-fp32 oracle (its own classes): BASELINE's 1e-4 on >= 99.5 % of them
    ref_fp32 = onp.brats_main_inr(p, vols, siren, 0, zmu, zsg, labels=lab, kind="siren", rows=(r0, r1))
    dpx = np.abs(got_band - ref_fp32)[..., :3].max(axis=-1)
    assert (dpx <= 1e-4).mean() >= 0.995, (dpx <= 1e-4).mean()
    print(f"C5 SIREN band: {seen.sum()} samples, argmax agreement {agree.mean():.5f}, worst tie gap "
          f"{(top2[:, 1] - top2[:, 0]).max() / scale if len(top2) else 0:.2e} of range; "
          f"frame: {a['queries']} queries chunked vs {a0['queries']} whole-ray, {a['live_samples']} live")
