"""INR forward on the GPU, with the reference's names and argument meaning.

Mirrors ``inr/inr/model.py`` as the viewer and notebooks use it (``from inr.model import
model_load, predict_volume``: inr/interactive.ipynb cell 5, inr/viewer/brats_viewer.py:261,293):

  model_load(npz_path, config_override=None) -> (params, config)        model.py:217-301
  build_input(coords, intensities, fourier_freqs) -> (B, 3+6K+M)        model.py:21-23
  apply_mlp(params, x) -> logits                                        model.py:43-50
  predict_volume(params, case_data, fourier_freqs, chunk) -> (pred,seg) model.py:119-141
  dice_score(pred, true, num_classes) / coverage_dice(pred, true)          model.py:144-161
  siren_apply(params, x, w0=30)                                         neumors_inr.ipynb:1165-1178

``params`` is the reference's list of ``{"W": [in,out], "b": [out]}`` (SIREN: dict ``l{i}`` ->
``{"w","b"}``).  All arithmetic runs in csrc/inr_mlp.hip (bf16 MFMA, fp32 accumulate, split-bf16
first layer); ``model_load`` is host-side file parsing.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import json
import pathlib
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .render import _ptr, _require_gpu, _stream_ptr

KIND_FOURIER_RELU, KIND_SIREN, KIND_RAW_RELU, KIND_RAW_SIREN = 0, 1, 2, 3


def model_load(npz_path, config_override: Optional[Dict[str, Any]] = None) -> Tuple[List[Dict[str, np.ndarray]], Dict[str, Any]]:
    """Checkpoint reader: ``{name}.npz`` + sidecar ``{name}_info.json`` (model.py:248-301).

    Accepts the final-checkpoint layout (key ``params`` holding a pickled list of W/b dicts,
    train.py:386-389 — needs ``allow_pickle`` and is therefore only honoured for files the caller
    trusts: pass ``config_override={"ALLOW_PICKLE": True}``) and the periodic-checkpoint layout
    (flat ``W_i`` / ``b_i`` arrays, train.py:216-223), which needs no pickle.
    """
    path = pathlib.Path(npz_path).expanduser().resolve()
    if not path.is_file():
        raise FileNotFoundError(f"no checkpoint archive at {path}")
    cfg_path = path.with_name(f"{path.stem}_info.json")
    if not cfg_path.is_file():
        raise FileNotFoundError(f"the sidecar {cfg_path.name} (training config) is missing beside {path.name}")
    allow_pickle = bool((config_override or {}).get("ALLOW_PICKLE", False))
    with np.load(str(path), allow_pickle=allow_pickle) as z:
        names = list(z.files)
        if names and all(n.startswith(("W_", "b_")) for n in names):
            count = sum(n.startswith("W_") for n in names)
            params = [{"W": np.asarray(z[f"W_{i}"]), "b": np.asarray(z[f"b_{i}"])} for i in range(count)]
        elif "params" in names or len(names) == 1:
            key = "params" if "params" in names else names[0]
            try:
                arr = z[key]
            except ValueError as exc:
                raise ValueError(f"{path} stores pickled params; pass config_override={{'ALLOW_PICKLE': True}} "
                                 "only for checkpoints you trust") from exc
            if arr.dtype == object:
                if arr.ndim != 0 and arr.size != 1:
                    raise ValueError(f"{path}: entry '{key}' holds {arr.size} pickled objects (shape {arr.shape}), "
                                     "a checkpoint stores exactly one parameter list")
                params = arr.item()
            else:
                params = arr
        else:
            raise KeyError(f"{path} has neither a 'params' entry nor W_i/b_i arrays (entries: {names})")
    config = json.loads(cfg_path.read_text())
    if config_override is not None:
        config = {**config, **{k: v for k, v in config_override.items() if k != "ALLOW_PICKLE"}}
    return params, config


def fourier_features(coords, k: int) -> torch.Tensor:
    """model.py:11-18: per axis [sin(pi 1 c) .. sin(pi k c), cos(pi 1 c) .. cos(pi k c)] -> (B, 6k)."""
    dev = _require_gpu()
    c = _dev_f32(coords, dev)
    freqs = torch.arange(1, k + 1, device=dev, dtype=torch.float32)
    ang = c[..., None] * freqs[None, None, :] * np.float32(np.pi)
    return torch.cat([torch.sin(ang), torch.cos(ang)], dim=-1).reshape(c.shape[0], -1)


@dataclass
class PackedMLP:
    """Device-resident network: permuted bf16 weight fragments + padded fp32 biases."""
    desc: _lib.InrDesc
    weights: torch.Tensor
    biases: torch.Tensor
    in_dim: int
    out_dim: int


def with_flags(net: PackedMLP, no_weight_stationary: bool = False, no_refine: bool = False, mark_only: bool = False,
               tie_sigmas: float = 0.0) -> PackedMLP:
    """The same packed network (shared device buffers) with other ``MrirtInrDesc.flags`` / ``tieSigmas``: the A/B switches
    of measurements and tests — the streaming kernel instead of the weight-stationary one, the bf16 pass without near-tie
    marking and second pass, marking without the second pass, another mark width.  (ABI 2 read these from the process
    environment at every launch.)"""
    d = _lib.InrDesc()
    C.memmove(C.byref(d), C.byref(net.desc), C.sizeof(d))
    d.flags = ((_lib.INR_NO_WEIGHT_STATIONARY if no_weight_stationary else 0) | (_lib.INR_NO_REFINE if no_refine else 0)
               | (_lib.INR_MARK_ONLY if mark_only else 0))
    d.tieSigmas = float(tie_sigmas)
    return PackedMLP(d, net.weights, net.biases, net.in_dim, net.out_dim)


def _layers(params) -> List[Tuple[np.ndarray, np.ndarray]]:
    if isinstance(params, dict):                         # SIREN notebook layout: l0, l1, ...
        return [(np.asarray(params[f"l{i}"]["w"], np.float32), np.asarray(params[f"l{i}"]["b"], np.float32))
                for i in range(len(params))]
    return [(np.asarray(p["W"], np.float32), np.asarray(p["b"], np.float32)) for p in params]


def pack_mlp(params, kind: int, fourier_freqs: int = 0, num_mods: int = 0, w0: float = 30.0) -> PackedMLP:
    dev = _require_gpu()
    layers = _layers(params)
    if len(layers) < 2:
        raise ValueError("the MLP kernel needs at least one hidden layer")
    hidden = layers[0][0].shape[1]
    for i, (W, b) in enumerate(layers):
        want_in = layers[0][0].shape[0] if i == 0 else hidden
        want_out = hidden if i + 1 < len(layers) else W.shape[1]
        if W.shape != (want_in, want_out) or b.shape != (want_out,):
            raise ValueError(f"layer {i}: W {W.shape} / b {b.shape}; the kernel needs equal hidden widths ({hidden})")
    d = _lib.InrDesc()
    d.kind, d.numLayers, d.inDim, d.outDim, d.hidden = kind, len(layers), layers[0][0].shape[0], layers[-1][0].shape[1], hidden
    d.fourierFreqs, d.numMods, d.w0 = int(fourier_freqs), int(num_mods), float(w0)
    nbytes = int(_lib.lib().mrirt_inr_pack_bytes(C.byref(d)))
    if nbytes <= 0:
        raise ValueError(f"unsupported network shape: in {d.inDim}, hidden {hidden} x {len(layers) - 1}, out {d.outDim} "
                         "(hidden in {32,64,128,256}, in <= 128, out <= 16, <= 8 layers)")
    w_flat = torch.from_numpy(np.concatenate([W.reshape(-1) for W, _ in layers])).to(dev)
    bias = np.concatenate([np.pad(b, (0, (-b.size) % 32)) for _, b in layers]).astype(np.float32)
    packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    biases = torch.from_numpy(bias).to(dev)
    d.weights, d.biases = packed.data_ptr(), biases.data_ptr()
    _lib.check(_lib.lib().mrirt_inr_pack_weights(C.byref(d), _ptr(w_flat), _ptr(packed), _stream_ptr(None)),
               "mrirt_inr_pack_weights")
    torch.cuda.current_stream().synchronize()           # w_flat may be freed after this returns
    return PackedMLP(d, packed, biases, int(d.inDim), int(d.outDim))


def _forward(net: PackedMLP, coords, feats, n: int, want_logits: bool, want_argmax: bool, refined: bool = False):
    dev = net.weights.device
    logits = torch.empty((n, net.out_dim), dtype=torch.float32, device=dev) if want_logits else None
    arg = torch.empty(n, dtype=torch.int16, device=dev) if want_argmax else None
    fn = _lib.lib().mrirt_inr_forward_refined if refined else _lib.lib().mrirt_inr_forward
    rc = fn(C.byref(net.desc), _ptr(coords), _ptr(feats), n, _ptr(logits), _ptr(arg), _stream_ptr(None))
    _lib.check(rc, "mrirt_inr_forward_refined" if refined else "mrirt_inr_forward")
    return logits, arg


def classify(net: PackedMLP, coords, feats, refined: bool = False, want_logits: bool = False):
    """argmax class (int16) of every point — the bf16 MFMA pass, with the points whose two largest logits are within
    the network's calibrated bf16 error re-evaluated in split bf16 (``mrirt_inr_forward``); ``refined=True`` evaluates
    every point in split bf16 (``mrirt_inr_forward_refined``).  Raw-x networks (KIND_RAW_*) take ``coords=None``."""
    dev = _require_gpu()
    f = _dev_f32(feats, dev)
    c = _dev_f32(coords, dev) if coords is not None else None
    logits, arg = _forward(net, c, f, f.shape[0], want_logits, True, refined)
    return (arg, logits) if want_logits else arg


def calibration(net: PackedMLP) -> Dict[str, float]:
    """The calibration record ``mrirt_inr_pack_weights`` measured for this network: rms / max difference between the
    bf16 pass's logits and the split-bf16 pass's over 8192 pseudo-random inputs, and the largest |logit| seen."""
    tail = net.weights[-1024:].view(torch.float32)[:4].cpu().numpy()
    return dict(rms_error=float(tail[0]), max_logit=float(tail[1]), max_error=float(tail[2]), points=int(tail[3]))


def _dev_f32(x, dev):
    return torch.as_tensor(np.asarray(x, dtype=np.float32) if not isinstance(x, torch.Tensor) else x,
                           dtype=torch.float32).to(dev).contiguous()


def build_input(coords, intensities, fourier_freqs: int) -> torch.Tensor:
    """(B,3) coords in [-1,1], (B,M) intensities -> (B, 3+6K+M) input matrix in the reference's
    feature order.  Provided for API parity; ``inr_forward`` builds the same features in-kernel."""
    dev = _require_gpu()
    c, f = _dev_f32(coords, dev), _dev_f32(intensities, dev)
    return torch.cat([c, fourier_features(c, fourier_freqs), f], dim=-1)


def inr_forward(params, coords, feats, fourier_freqs: int, net: Optional[PackedMLP] = None):
    """logits (B, classes) fp32 for points: Fourier features + ReLU MLP, fused."""
    dev = _require_gpu()
    f = _dev_f32(feats, dev)
    net = net or pack_mlp(params, KIND_FOURIER_RELU, fourier_freqs, f.shape[1])
    c = _dev_f32(coords, dev)
    return _forward(net, c, f, c.shape[0], True, False)[0]


def apply_mlp(params, x, net: Optional[PackedMLP] = None) -> torch.Tensor:
    """model.py:43-50 on an already-built input matrix x (B, in_dim)."""
    dev = _require_gpu()
    xx = _dev_f32(x, dev)
    net = net or pack_mlp(params, KIND_RAW_RELU)
    return _forward(net, None, xx, xx.shape[0], True, False)[0]


def siren_apply(params, x, w0: float = 30.0, net: Optional[PackedMLP] = None) -> torch.Tensor:
    dev = _require_gpu()
    xx = _dev_f32(x, dev)
    net = net or pack_mlp(params, KIND_RAW_SIREN, w0=w0)
    return _forward(net, None, xx, xx.shape[0], True, False)[0]


def predict_volume(params, case_data: Dict[str, Any], fourier_freqs: int, chunk: int = 200000,
                   net: Optional[PackedMLP] = None):
    """model.py:119-141: argmax class per voxel of ``case_data["mods"]`` (M,H,W,D) -> int16 (H,W,D).
    ``chunk`` is accepted for signature parity; the kernel streams the whole volume in one launch."""
    dev = _require_gpu()
    mods = _dev_f32(case_data["mods"], dev)
    M, H, W, D = mods.shape
    net = net or pack_mlp(params, KIND_FOURIER_RELU, fourier_freqs, M)
    pred = torch.empty((H, W, D), dtype=torch.int16, device=dev)
    hwd = (C.c_uint32 * 3)(H, W, D)
    rc = _lib.lib().mrirt_inr_predict_volume(C.byref(net.desc), _ptr(mods), hwd, _ptr(pred), _stream_ptr(None))
    _lib.check(rc, "mrirt_inr_predict_volume")
    return pred, case_data["seg"]


def _label_counts(pred, true, num_classes: int):
    """Joint histogram of two label volumes: counts[p, t] for labels below ``num_classes`` (others in an overflow bin).
    NumPy arrays stay on the host; torch tensors are counted where they live (one bincount, one small read-back)."""
    n = int(num_classes)
    if isinstance(pred, torch.Tensor) or isinstance(true, torch.Tensor):
        dev = pred.device if isinstance(pred, torch.Tensor) else true.device
        p = torch.as_tensor(pred, device=dev).reshape(-1).to(torch.int64)
        t = torch.as_tensor(true, device=dev).reshape(-1).to(torch.int64)
        if p.numel() != t.numel():
            raise ValueError("dice: prediction and ground truth differ in size")
        p = torch.where((p < 0) | (p >= n), torch.full_like(p, n), p)
        t = torch.where((t < 0) | (t >= n), torch.full_like(t, n), t)
        return torch.bincount(p * (n + 1) + t, minlength=(n + 1) * (n + 1)).reshape(n + 1, n + 1).cpu().numpy()
    p = np.asarray(pred).reshape(-1).astype(np.int64)
    t = np.asarray(true).reshape(-1).astype(np.int64)
    if p.size != t.size:
        raise ValueError("dice: prediction and ground truth differ in size")
    p = np.where((p < 0) | (p >= n), n, p)
    t = np.where((t < 0) | (t >= n), n, t)
    return np.bincount(p * (n + 1) + t, minlength=(n + 1) * (n + 1)).reshape(n + 1, n + 1)


def dice_score(pred, true, num_classes: int) -> Dict[int, float]:
    """Per-class Dice of two label volumes — the third name inr/interactive.ipynb imports from inr.model next to
    model_load and predict_volume (inr/inr/model.py:144-153): ``{c: (2 |P_c & T_c| + 1e-6) / (|P_c| + |T_c| + 1e-6)}``,
    NaN for a class that occurs in neither volume.  Accepts what ``predict_volume`` returns (a device tensor) as is."""
    h = _label_counts(pred, true, num_classes)
    out: Dict[int, float] = {}
    for c in range(int(num_classes)):
        denom = int(h[c, :].sum()) + int(h[:, c].sum())
        out[c] = (2 * int(h[c, c]) + 1e-6) / (denom + 1e-6) if denom > 0 else float("nan")
    return out


def coverage_dice(pred, true) -> float:
    """Dice of the foregrounds (label > 0) of two label volumes (inr/inr/model.py:156-161); 0.0 when both are empty."""
    pf = (pred > 0)
    tf = (true > 0)
    h = _label_counts(pf.to(torch.int64) if isinstance(pf, torch.Tensor) else np.asarray(pf, dtype=np.int64),
                      tf.to(torch.int64) if isinstance(tf, torch.Tensor) else np.asarray(tf, dtype=np.int64), 2)
    denom = int(h[1, :].sum()) + int(h[:, 1].sum())
    return (2 * int(h[1, 1]) + 1e-6) / (denom + 1e-6) if denom > 0 else 0.0


_C5_SCRATCH: dict = {}


def render_brats_inr(params, intensities, net: PackedMLP, zmu, zsigma, labels=None, out=None, ext=None,
                     return_aux: bool = False, chunk_steps: Optional[int] = None, one_pass: bool = False):
    """BASELINE config 5 (build-defined): K1 with the prediction overlay's label taken
    from the MLP evaluated AT every march sample (normalised sample coordinates + the four
    trilinear-sampled, z-scored modalities) instead of ``sampleLabel(gPreds)``.

    ``net`` is a packed network over 4 modalities: the reference's Fourier/ReLU MLP
    (``pack_mlp(params, KIND_FOURIER_RELU, K, 4)``, inr/inr/model.py:11-50) or the notebook's SIREN
    (``pack_mlp(params, KIND_SIREN, 0, 4)``: x = (coords, modalities), neumors_inr.ipynb:853-899,1165-1178).
    ``zmu``/``zsigma`` are the per-modality z-score constants (brats_viewer.py:281-287).

    Default: ``mrirt_render_brats_inr`` — the march advances ``chunk_steps`` per pass and only rays that are still
    alive (t < t1, T > 0.01) have their next samples classified ("all live sample points"); no host
    synchronisation inside the frame.  ``chunk_steps=None`` picks the pass length from the scene: where the intensity
    channel alone cannot take any ray to T <= ert (intensityAlpha x box diagonal <= -ln ert — the reference viewer's whole
    slider range) no sample is classified in vain whatever the pass length, and 96-step passes are the
    fastest (fewer launches and refinement rounds: 10.13 / 9.88 / 9.68 / 9.72 / 9.76 ms per frame with 32 / 64 / 96 / 128 / 256
    steps on the config-5 scene, round 4); otherwise 32
    (a terminating ray wastes at most 31 samples; 8.9 against 9.2 ms on the dense preset).  Any value gives the same bits.  ``one_pass=True`` is the three-pass form over whole rays (count every
    sample in [t0,t1) -> emit -> ONE batched forward -> composite with the class stream); both give the same
    bits, and ``return_aux`` of the one-pass form exposes the emitted inputs and classes for the layered tests.
    """
    from .render import Grid, _alloc_out, _bind_brats
    dev = _require_gpu()
    if isinstance(intensities, Grid):                    # one "mod4" grid (upload_mod4) carries all four modalities
        if intensities.layout != "mod4":
            raise ValueError("a single grid must be the 'mod4' grid of upload_mod4; per-modality grids go in a list of four")
        if one_pass:
            raise ValueError("the whole-ray three-pass form marches with the K1 kernels: bind per-modality grids")
        intensities = [intensities] * 4
    if int(params["showPred"]) == 0:
        raise ValueError("render_brats_inr draws the prediction overlay: set gParams.showPred")
    if net.desc.kind not in (KIND_FOURIER_RELU, KIND_SIREN) or net.desc.numMods != 4:
        raise ValueError("render_brats_inr needs a Fourier/ReLU or SIREN network over 4 modalities "
                         "(pack_mlp(params, KIND_FOURIER_RELU, K, 4) / pack_mlp(params, KIND_SIREN, 0, 4))")
    P, E, vols, lab, _ = _bind_brats(params, intensities, labels, None, ext, dev, pred_stream=True)
    if any(v is None for v in vols):
        raise ValueError("the MLP reads all four modalities: bind gIntensity0..3")
    if E.tileSize > 0:
        raise ValueError("render_brats_inr renders whole frames")
    w, h = int(P.imageSize[0]), int(P.imageSize[1])
    lib, s = _lib.lib(), _stream_ptr(None)
    vp = (C.c_void_p * 4)(*[C.c_void_p(t.data_ptr()) for t in vols])
    mu = (C.c_float * 4)(*[float(np.float32(v)) for v in zmu])
    sg = (C.c_float * 4)(*[float(np.float32(v)) for v in zsigma])
    if chunk_steps is None:
        diag = float(np.sqrt(sum((float(P.voxelSize[k]) * int(P.dims[k])) ** 2 for k in range(3))))
        ert = float(E.ertThreshold) if int(E.ertOverride) else 0.01
        chunk_steps = 96 if ert > 0.0 and float(P.intensityAlpha) * diag <= -np.log(ert) else 32
    if not one_pass:
        nbytes = int(lib.mrirt_brats_inr_scratch_bytes(C.byref(P), int(chunk_steps)))
        if nbytes <= 0:
            raise ValueError(f"chunk_steps={chunk_steps}: unsupported (1..4096, image x chunk < 2^32 samples per pass)")
        key = (dev.index, torch.cuda.current_stream().cuda_stream)
        scratch = _C5_SCRATCH.get(key)                    # one scratch per (device, stream): reused frame to frame
        if scratch is None or scratch.numel() < nbytes:
            scratch = _C5_SCRATCH[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        o, pitch = _alloc_out(w, h, E, dev, out)
        st = torch.zeros(3, dtype=torch.int64, device=dev) if return_aux else None
        _lib.check(lib.mrirt_render_brats_inr(C.byref(P), C.byref(E), vp, _ptr(lab), C.byref(net.desc), mu, sg,
                                              int(chunk_steps), _ptr(scratch), scratch.numel(), _ptr(o), pitch,
                                              _ptr(st), s), "mrirt_render_brats_inr")
        if return_aux:
            c = st.cpu()
            return o, dict(live_samples=int(c[0]), shaded_samples=int(c[1]), queries=int(c[2]), chunk_steps=int(chunk_steps))
        return o
    counts = torch.empty(h * w, dtype=torch.int32, device=dev)
    _lib.check(lib.mrirt_brats_sample_counts(C.byref(P), C.byref(E), _ptr(counts), s), "mrirt_brats_sample_counts")
    ends = torch.cumsum(counts.to(torch.int64), 0)
    offsets = (ends - counts).contiguous()
    total = int(ends[-1])
    coords = torch.empty((max(total, 1), 3), dtype=torch.float32, device=dev)
    feats = torch.empty((max(total, 1), 4), dtype=torch.float32, device=dev)
    classes = torch.zeros(max(total, 1), dtype=torch.int16, device=dev)
    _lib.check(lib.mrirt_brats_emit_samples(C.byref(P), C.byref(E), vp, mu, sg, _ptr(offsets), _ptr(coords), _ptr(feats), s),
               "mrirt_brats_emit_samples")
    if total:
        _lib.check(lib.mrirt_inr_forward(C.byref(net.desc), _ptr(coords), _ptr(feats), total, None, _ptr(classes), s),
                   "mrirt_inr_forward")
    o, pitch = _alloc_out(w, h, E, dev, out)
    st = torch.zeros(2, dtype=torch.int64, device=dev) if return_aux else None
    _lib.check(lib.mrirt_render_brats_stream(C.byref(P), C.byref(E), vp, _ptr(lab), _ptr(classes), _ptr(offsets),
                                             _ptr(o), pitch, _ptr(st), s), "mrirt_render_brats_stream")
    if return_aux:
        c = st.cpu()
        return o, dict(queries=total, live_samples=int(c[0]), shaded_samples=int(c[1]), classes=classes, offsets=offsets, coords=coords,
                       feats=feats, counts=counts)
    return o


def labels_for_viewer(pred_hwd: torch.Tensor) -> torch.Tensor:
    """pred (H,W,D) -> the viewer's x-fastest uint32 label buffer (brats_viewer.py:297-299)."""
    return pred_hwd.permute(2, 1, 0).reshape(-1).to(torch.int32).contiguous()
