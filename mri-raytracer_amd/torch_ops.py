"""``torch.ops.mrirt.*`` — the C ABI of libmrirt.so registered as PyTorch custom operators.

Device tensors in, a device tensor out, launched on the current HIP stream, no implicit
synchronisation.  The operators are thin: parameter blocks travel as
CPU ``uint8`` tensors holding the C structs of include/mrirt.h byte for byte (``pack_brats_params``
& co. build them from the reference's ``gParams`` dicts), every size check the C side cannot make
(it only sees pointers) is made here, and the launch itself is the same ``mrirt_*`` entry point the
rest of the package calls.  There is no CPU implementation: the only other registration is the
shape function ("fake" kernel) that ``torch.compile`` / meta tensors need.  The render operators also exist as a
C++ extension (``load_native()`` -> ``torch.ops.mrirt_native.*``, csrc/torch_binding.cpp).

    blob = mrirt.torch_ops.pack_brats_params(gparams)
    ext  = mrirt.torch_ops.pack_render_ext({"layout": "vg", **mrirt.synth.SHADE_EXT})
    img  = torch.ops.mrirt.render_brats(blob, ext, grid.data, None, None, None, None, None)
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Mapping, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .params import brats_params, render_ext, sdf_params, volume_params


def load_native():
    """``torch.ops.mrirt_native`` — the same render operators registered from C++ (csrc/torch_binding.cpp, a
    PyTorch-ROCm C++ extension linked to libmrirt.so; schemas as ``torch.ops.mrirt.render_*`` below).  Raises if
    libmrirt_torch.so has not been built (``__graft_entry__.build()``): like the HIP library, it has no fallback."""
    if not _lib.TORCH_SO_PATH.exists():
        raise ImportError(f"{_lib.TORCH_SO_PATH} is missing — build it with `python -c \"import __graft_entry__ as g; g.build()\"`.")
    _lib.lib()                                    # libmrirt.so first: the operator library links against it
    torch.ops.load_library(str(_lib.TORCH_SO_PATH))
    return torch.ops.mrirt_native


# --- parameter blocks <-> uint8 tensors ---------------------------------------------------------
def _blob(struct: C.Structure) -> torch.Tensor:
    return torch.from_numpy(np.frombuffer(bytes(struct), dtype=np.uint8).copy())


def _unblob(t: torch.Tensor, ctype):
    if t.device.type != "cpu" or t.dtype != torch.uint8 or t.numel() != C.sizeof(ctype):
        raise TypeError(f"expected a CPU uint8 tensor of {C.sizeof(ctype)} bytes ({ctype.__name__})")
    return ctype.from_buffer_copy(t.contiguous().numpy().tobytes())


def pack_brats_params(p: Mapping[str, Any]) -> torch.Tensor:
    """gParams dict of inr/viewer/brats_viewer.py:405-426 -> MrirtBratsParams bytes."""
    return _blob(brats_params(p))


def pack_render_ext(ext: Optional[Mapping[str, Any]] = None) -> torch.Tensor:
    return _blob(render_ext(ext))


def pack_volume_params(p: Mapping[str, Any]) -> torch.Tensor:
    """gParams dict of scripts/volumeRendering/app.py:331-345 -> MrirtVolumeParams bytes."""
    return _blob(volume_params(p))


def pack_sdf_params(p: Mapping[str, Any], eye, U, V, W) -> torch.Tensor:
    """gParams + gEye/gU/gV/gW of scripts/raymarch/app.py:199-221 -> MrirtSdfParams bytes."""
    return _blob(sdf_params(p, eye, U, V, W))


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


def _one_device(named: Sequence) -> torch.device:
    """The GPU every device tensor of a call lives on (raises when they disagree).  The launch then happens with that
    GPU current (``torch.cuda.device``): the stream is ITS current stream and the output is allocated on it — with
    device 0 current and the grids on device 1 the kernel would otherwise be enqueued on device 0 with device-1
    pointers."""
    dev = None
    for what, t in named:
        if t is None:
            continue
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise ValueError(f"{what} is on {t.device} but another operand is on {dev}: every device tensor of one call "
                             "must live on the same GPU")
    if dev is None:
        raise ValueError("no device tensor bound")
    return dev


def _dev_flat(t: Optional[torch.Tensor], dtype, what: str) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise TypeError(f"{what}: expected a contiguous device {dtype} tensor")
    return t


def _out_shape(width: int, height: int, E: _lib.RenderExt):
    if E.tileSize > 0:
        tiles_x = (width + E.tileSize - 1) // E.tileSize
        tiles_y = (height + E.tileSize - 1) // E.tileSize
        total = tiles_x * tiles_y
        local = (total - E.tileRank + E.tileWorld - 1) // E.tileWorld if E.tileRank < total else 0
        return (local, int(E.tileSize), int(E.tileSize), 4)
    return (height, width, 4)


def _grid_need(dims: Sequence[int], layout: int) -> int:
    l = _lib.lib()
    d = (C.c_uint32 * 3)(*dims)
    if layout == _lib.LAYOUT_LINEAR:
        return dims[0] * dims[1] * dims[2]
    if layout == _lib.LAYOUT_BRICK:
        return int(l.mrirt_brick_elems(d))
    if layout == _lib.LAYOUT_VGA:                    # three axis-flat copies of the float4 voxels
        return 4 * int(l.mrirt_vga_elems(d))
    if layout == _lib.LAYOUT_LABCELL:                # both overlays' corner labels per cell: 8 bytes per element
        return 2 * int(l.mrirt_vec4_elems(d))
    return 4 * int(l.mrirt_vec4_elems(d))


# --- K1 -----------------------------------------------------------------------------------------
@torch.library.custom_op("mrirt::render_brats", mutates_args=())
def render_brats(params: torch.Tensor, ext: torch.Tensor, vol0: Optional[torch.Tensor], vol1: Optional[torch.Tensor],
                 vol2: Optional[torch.Tensor], vol3: Optional[torch.Tensor], labels: Optional[torch.Tensor],
                 preds: Optional[torch.Tensor]) -> torch.Tensor:
    """brats_main (inr/viewer/brats_rt.slang:85-168) through mrirt_render_brats_ex."""
    P, E = _unblob(params, _lib.BratsParams), _unblob(ext, _lib.RenderExt)
    dims = [int(v) for v in P.dims]
    vols = [_dev_flat(v, torch.float32, f"gIntensity{m}") for m, v in enumerate((vol0, vol1, vol2, vol3))]
    lab = _dev_flat(labels, torch.int32, "gLabels")
    prd = _dev_flat(preds, torch.int32, "gPreds")
    need, lneed = _grid_need(dims, E.layout), _grid_need(dims, E.labelLayout)
    mod4 = E.layout == _lib.LAYOUT_MOD4                  # vol0 = the float4 grid of all four modalities (vol1..3 are ignored)
    for m, v in enumerate(vols):
        if (m == 0 if mod4 else P.volEnabled[m] != 0) and (v is None or v.numel() < need):
            raise ValueError(f"gIntensity{m} is {'the MOD4 grid' if mod4 else 'enabled'} but holds {0 if v is None else v.numel()} < {need} elements")
    if P.showSeg != 0 and (lab is None or lab.numel() < lneed):
        raise ValueError("showSeg is set but gLabels is missing or too small")
    cells = E.labelLayout == _lib.LAYOUT_LABCELL     # gLabels carries both grids, gPreds is ignored
    if P.showPred != 0 and ((lab if cells else prd) is None or (lab if cells else prd).numel() < lneed):
        raise ValueError("showPred is set but gPreds (or the label-cell grid) is missing or too small")
    dev = _one_device([(f"gIntensity{m}", v) for m, v in enumerate(vols)] + [("gLabels", lab), ("gPreds", prd)])
    dt = torch.float16 if E.outFormat == _lib.OUT_RGBA16F else torch.float32
    vp = (C.c_void_p * 4)(*[C.c_void_p(v.data_ptr()) if v is not None else None for v in vols])
    with torch.cuda.device(dev):
        out = torch.empty(_out_shape(int(P.imageSize[0]), int(P.imageSize[1]), E), dtype=dt, device=dev)
        rc = _lib.lib().mrirt_render_brats_ex(C.byref(P), C.byref(E), vp, _ptr(lab), _ptr(prd), _ptr(out),
                                              int(P.imageSize[0]), None, _stream())
    _lib.check(rc, "mrirt_render_brats_ex")
    return out


@render_brats.register_fake
def _(params, ext, vol0, vol1, vol2, vol3, labels, preds):
    P, E = _unblob(params, _lib.BratsParams), _unblob(ext, _lib.RenderExt)
    dev = next(v.device for v in (vol0, vol1, vol2, vol3) if v is not None)
    dt = torch.float16 if E.outFormat == _lib.OUT_RGBA16F else torch.float32
    return torch.empty(_out_shape(int(P.imageSize[0]), int(P.imageSize[1]), E), dtype=dt, device=dev)


# --- K2 -----------------------------------------------------------------------------------------
_VOX_DTYPE = {_lib.VOX_U32X4: torch.int32, _lib.VOX_U8: torch.uint8, _lib.VOX_F32: torch.float32}


@torch.library.custom_op("mrirt::render_volume", mutates_args=())
def render_volume(params: torch.Tensor, ext: torch.Tensor, volume: torch.Tensor, mode: int) -> torch.Tensor:
    """volume_cs (scripts/volumeRendering/volume_render.slang:104-148) through mrirt_render_volume;
    ``mode``: 0 one-u32-per-voxel (the reference's upload), 1 bytes, 2 fp32."""
    P, E = _unblob(params, _lib.VolumeParams), _unblob(ext, _lib.RenderExt)
    if mode not in _VOX_DTYPE:
        raise ValueError("mode must be 0 (u32x4), 1 (u8) or 2 (f32)")
    vol = _dev_flat(volume, _VOX_DTYPE[mode], "gVolumeU8")
    nvox = int(P.volDim[0]) * int(P.volDim[1]) * int(P.volDim[2])
    if vol.numel() < nvox:
        raise ValueError(f"gVolumeU8 holds {vol.numel()} < {nvox} voxels")
    dt = torch.float16 if E.outFormat == _lib.OUT_RGBA16F else torch.float32
    with torch.cuda.device(vol.device):
        out = torch.empty(_out_shape(int(P.imageSize[0]), int(P.imageSize[1]), E), dtype=dt, device=vol.device)
        rc = _lib.lib().mrirt_render_volume(C.byref(P), C.byref(E), _ptr(vol), int(mode), _ptr(out), int(P.imageSize[0]),
                                            None, _stream())
    _lib.check(rc, "mrirt_render_volume")
    return out


@render_volume.register_fake
def _(params, ext, volume, mode):
    P, E = _unblob(params, _lib.VolumeParams), _unblob(ext, _lib.RenderExt)
    dt = torch.float16 if E.outFormat == _lib.OUT_RGBA16F else torch.float32
    return torch.empty(_out_shape(int(P.imageSize[0]), int(P.imageSize[1]), E), dtype=dt, device=volume.device)


# --- K3 -----------------------------------------------------------------------------------------
@torch.library.custom_op("mrirt::render_sdf", mutates_args=())
def render_sdf(params: torch.Tensor, width: int, height: int, like: torch.Tensor) -> torch.Tensor:
    """raymarch_cs (scripts/raymarch/raymarch.slang:60-99); ``like`` only names the device."""
    P = _unblob(params, _lib.SdfParams)
    if not like.is_cuda:
        raise TypeError("like: expected a device tensor")
    with torch.cuda.device(like.device):
        out = torch.empty((height, width, 4), dtype=torch.float32, device=like.device)
        rc = _lib.lib().mrirt_render_sdf(C.byref(P), int(width), int(height), _ptr(out), int(width), _stream())
    _lib.check(rc, "mrirt_render_sdf")
    return out


@render_sdf.register_fake
def _(params, width, height, like):
    return torch.empty((height, width, 4), dtype=torch.float32, device=like.device)


# --- INR ----------------------------------------------------------------------------------------
@torch.library.custom_op("mrirt::inr_forward", mutates_args=())
def inr_forward(weights: torch.Tensor, biases: torch.Tensor, kind: int, num_layers: int, in_dim: int, out_dim: int,
                hidden: int, fourier_freqs: int, num_mods: int, w0: float, coords: Optional[torch.Tensor],
                feats: Optional[torch.Tensor], n: int) -> torch.Tensor:
    """Logits [n, out_dim] of the packed MLP (``mrirt.inr.pack_mlp(...).weights / .biases``) for n points:
    inr/inr/model.py:21-50 (kind 0), notebooks/neumors_inr.ipynb:1165-1178 (kind 1), or the raw-input
    forms (kinds 2/3: ``feats`` is the [n, in_dim] input matrix)."""
    d = _lib.InrDesc()
    d.kind, d.numLayers, d.inDim, d.outDim, d.hidden = kind, num_layers, in_dim, out_dim, hidden
    d.fourierFreqs, d.numMods, d.w0 = fourier_freqs, num_mods, w0
    need = int(_lib.lib().mrirt_inr_pack_bytes(C.byref(d)))
    if need <= 0:
        raise ValueError("unsupported network shape")
    if not weights.is_cuda or weights.dtype != torch.uint8 or weights.numel() < need:
        raise TypeError(f"weights: expected the {need}-byte packed image of mrirt_inr_pack_weights on the device")
    nb = (num_layers - 1) * hidden + ((out_dim + 31) // 32) * 32
    bz = _dev_flat(biases, torch.float32, "biases")
    if bz.numel() < nb:
        raise ValueError(f"biases holds {bz.numel()} < {nb} floats (each layer padded to a multiple of 32)")
    co = _dev_flat(coords, torch.float32, "coords")
    fe = _dev_flat(feats, torch.float32, "feats")
    if kind < 2 and (co is None or co.numel() < 3 * n):
        raise ValueError("coords must hold [n, 3] floats")
    width = in_dim if kind >= 2 else num_mods
    if width > 0 and (fe is None or fe.numel() < width * n):
        raise ValueError(f"feats must hold [n, {width}] floats")
    d.weights, d.biases = weights.data_ptr(), bz.data_ptr()
    dev = _one_device([("weights", weights), ("biases", bz), ("coords", co), ("feats", fe)])
    with torch.cuda.device(dev):
        out = torch.empty((n, out_dim), dtype=torch.float32, device=dev)
        rc = _lib.lib().mrirt_inr_forward(C.byref(d), _ptr(co), _ptr(fe), int(n), _ptr(out), None, _stream())
    _lib.check(rc, "mrirt_inr_forward")
    return out


@inr_forward.register_fake
def _(weights, biases, kind, num_layers, in_dim, out_dim, hidden, fourier_freqs, num_mods, w0, coords, feats, n):
    return torch.empty((n, out_dim), dtype=torch.float32, device=weights.device)
