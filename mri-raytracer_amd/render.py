"""Render entry points over the C ABI: render_brats / render_volume_u8 / render_sdf.

torch is used for device memory and the current HIP stream only; every pixel is produced by
the hand-written gfx950 kernels in csrc/ (reached through libmrirt.so).  There is no CPU path:
without a GPU (or without the built library) these functions raise.
"""
from __future__ import annotations

import collections
import contextlib
import ctypes as C
from dataclasses import dataclass
from typing import Any, Dict, Mapping, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib
from .params import brats_params, render_ext, sdf_params, volume_params

ArrayLike = Union[np.ndarray, torch.Tensor]


def _require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("mrirt: no HIP device visible — the ray-marcher has no CPU fallback "
                           "(the CPU restatement under oracle/ is test infrastructure only)")
    return torch.device("cuda", torch.cuda.current_device())


def _stream_ptr(stream) -> C.c_void_p:
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream if hasattr(s, "cuda_stream") else int(s))


def _on_stream(stream):
    """Context that makes `stream` torch's current stream (allocations and host->device copies follow it);
    a no-op for the default (None) and for raw hipStream_t integers, which torch cannot adopt without owning."""
    if stream is not None and isinstance(stream, torch.cuda.Stream):
        return torch.cuda.stream(stream)
    return contextlib.nullcontext()


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(t.data_ptr() if t is not None else None)


@dataclass
class Grid:
    """A device-resident voxel grid: the analogue of the reference's StructuredBuffer created by
    ``device.create_buffer`` + ``copy_from_numpy`` (inr/viewer/brats_viewer.py:219-230)."""
    data: torch.Tensor            # 1-D, device
    dims: Tuple[int, int, int]    # (X, Y, Z), x fastest in the linear layout
    layout: str = "linear"        # "linear" | "brick" | "vg" | "quad" | "vga"
    macro: Optional[torch.Tensor] = None   # per 8^3 macro cell: fp32 upper bound (intensities) / any-label word
                                           # (label grids) — what exact empty-space skipping tests (skip=True)
    macro2: Optional[torch.Tensor] = None  # "labcell" grids: the prediction grid's summary (macro = the ground truth's)

    @property
    def nbytes(self) -> int:
        return self.data.numel() * self.data.element_size()


def brick_elems(dims: Sequence[int]) -> int:
    d = (C.c_uint32 * 3)(*[int(v) for v in dims])
    return int(_lib.lib().mrirt_brick_elems(d))


def vec4_elems(dims: Sequence[int]) -> int:
    d = (C.c_uint32 * 3)(*[int(v) for v in dims])
    return int(_lib.lib().mrirt_vec4_elems(d))


def vga_elems(dims: Sequence[int]) -> int:
    """float4 elements of the three axis-flat copies of a "vga" grid together."""
    d = (C.c_uint32 * 3)(*[int(v) for v in dims])
    return int(_lib.lib().mrirt_vga_elems(d))


def upload_grid(linear: ArrayLike, dims: Sequence[int], layout: str = "brick", stream=None, macro: bool = True) -> Grid:
    """Upload a linear (x-fastest) fp32 / uint32 / uint8 grid and convert it on the device
    (csrc/grid_ops.hip) to ``layout``: "linear" (as is), "brick" (4x4x2 bricks), or for fp32
    intensities "vg" (float4 value + lattice gradient) / "quad" (float4 xy-neighbours) / "vga" (the "vg" voxels
    three times, in bricks one voxel thick along x, y, z: the march reads the copy that is flat along the face its
    ray packet entered through — 3x the memory, about half the cache lines per gather).
    Load-time, once per volume.  ``macro`` also builds the 8^3 macro-cell summary that
    ``render_brats(..., skip=True)`` needs (a few hundred KB)."""
    dev = _require_gpu()
    dims = tuple(int(v) for v in dims)
    t = torch.as_tensor(linear).reshape(-1)
    if t.dtype not in (torch.float32, torch.uint8, torch.int32, torch.uint32):
        if t.dtype in (torch.int64, torch.uint16, torch.int16):
            t = t.to(torch.int32)
        else:
            t = t.to(torch.float32)
    if t.numel() != dims[0] * dims[1] * dims[2]:
        raise ValueError(f"grid has {t.numel()} voxels, dims {dims} need {dims[0] * dims[1] * dims[2]}")
    t = t.to(dev).contiguous()
    macro = _build_macro(t, dims, stream) if macro else None
    if layout == "linear":
        return Grid(t, dims, "linear", macro)
    if layout in ("vg", "quad", "vga"):
        if t.dtype != torch.float32:
            raise TypeError(f"layout {layout!r} is for fp32 intensity grids, got {t.dtype}")
        out = torch.empty(4 * (vga_elems(dims) if layout == "vga" else vec4_elems(dims)), dtype=torch.float32, device=dev)
        d = (C.c_uint32 * 3)(*dims)
        code = {"vg": _lib.LAYOUT_VG, "quad": _lib.LAYOUT_QUAD, "vga": _lib.LAYOUT_VGA}[layout]
        _lib.check(_lib.lib().mrirt_build_vec4_grid(_ptr(t), _ptr(out), d, code, _stream_ptr(stream)),
                   "mrirt_build_vec4_grid")
        return Grid(out, dims, layout, macro)
    if layout != "brick":
        raise ValueError(f"unknown layout {layout!r}")
    out = torch.empty(brick_elems(dims), dtype=t.dtype, device=dev)
    d = (C.c_uint32 * 3)(*dims)
    _lib.check(_lib.lib().mrirt_brick_grid(_ptr(t), _ptr(out), d, t.element_size(), _stream_ptr(stream)),
               "mrirt_brick_grid")
    return Grid(out, dims, "brick", macro)


def upload_label_cells(seg: Optional[ArrayLike], pred: Optional[ArrayLike], dims: Sequence[int], stream=None, macro: bool = True) -> Grid:
    """Both label grids of the viewer (gLabels, gPreds: linear uint32, either may be None) as ONE "labcell" grid for QUAD
    intensity grids: per cell the corner labels of both as nibbles, in the QUAD grid's element order (include/mrirt.h,
    MRIRT_LAYOUT_LABCELL).  sampleLabel's rounded voxel is always one of the sample's cell corners and the shader draws labels
    1..7 only, so a sample takes one 8-byte gather at the offset its intensity taps already have instead of two nearest-voxel
    gathers — same bits.  Bind it as ``labels=`` (``preds`` is then ignored)."""
    dev = _require_gpu()
    dims = tuple(int(v) for v in dims)
    n = dims[0] * dims[1] * dims[2]
    lin, macros = [], []
    for g, what in ((seg, "seg"), (pred, "pred")):
        if g is None:
            lin.append(None); macros.append(None)
            continue
        t = _as_device_tensor(g, torch.int32, dev, what)
        if t.numel() != n:
            raise ValueError(f"{what} has {t.numel()} voxels, dims {dims} need {n}")
        lin.append(t)
        macros.append(_build_macro(t, dims, stream) if macro else None)
    out = torch.empty(2 * vec4_elems(dims), dtype=torch.int32, device=dev)
    d = (C.c_uint32 * 3)(*dims)
    _lib.check(_lib.lib().mrirt_build_label_cells(_ptr(lin[0]), _ptr(lin[1]), d, _ptr(out), _stream_ptr(stream)), "mrirt_build_label_cells")
    return Grid(out, dims, "labcell", macros[0], macros[1])


def _build_macro(t: torch.Tensor, dims, stream=None) -> Optional[torch.Tensor]:
    """Macro-cell summary of a LINEAR device grid (csrc/grid_ops.hip): fp32 -> upper bound of the trilinear
    fetch per 8^3 cell; int32/uint32 labels -> OR of the labels per cell; other dtypes: none."""
    d = (C.c_uint32 * 3)(*dims)
    cells = int(_lib.lib().mrirt_macro_cells(d))
    if t.dtype == torch.float32:
        out = torch.empty(cells, dtype=torch.float32, device=t.device)
        _lib.check(_lib.lib().mrirt_build_macro_max(_ptr(t), d, _ptr(out), _stream_ptr(stream)), "mrirt_build_macro_max")
        return out
    if t.dtype in (torch.int32, torch.uint32):
        out = torch.empty(cells, dtype=torch.int32, device=t.device)
        _lib.check(_lib.lib().mrirt_build_macro_labels(_ptr(t), d, _ptr(out), _stream_ptr(stream)), "mrirt_build_macro_labels")
        return out
    return None


def unbrick_grid(g: Grid, stream=None) -> torch.Tensor:
    if g.layout == "linear":
        return g.data
    out = torch.empty(g.dims[0] * g.dims[1] * g.dims[2], dtype=g.data.dtype, device=g.data.device)
    d = (C.c_uint32 * 3)(*g.dims)
    _lib.check(_lib.lib().mrirt_unbrick_grid(_ptr(g.data), _ptr(out), d, g.data.element_size(), _stream_ptr(stream)),
               "mrirt_unbrick_grid")
    return out


def _as_device_tensor(x, dtype, dev, what: str) -> Optional[torch.Tensor]:
    if x is None:
        return None
    if isinstance(x, Grid):
        x = x.data
    t = torch.as_tensor(x)
    if t.dtype != dtype:
        if dtype == torch.int32 and t.dtype in (torch.uint32, torch.int64, torch.uint8, torch.int16):
            t = t.to(torch.int64).to(torch.int32) if t.dtype == torch.uint32 else t.to(torch.int32)
        elif dtype == torch.float32 and t.is_floating_point():
            t = t.to(torch.float32)
        else:
            raise TypeError(f"{what}: expected {dtype}, got {t.dtype}")
    return t.to(dev).contiguous().reshape(-1)


def tiles_for_rank(width: int, height: int, tile: int, rank: int, world: int) -> int:
    return int(_lib.lib().mrirt_tiles_for_rank(width, height, tile, rank, world))


def _alloc_out(params_w: int, params_h: int, ext: _lib.RenderExt, dev, out):
    dt = torch.float16 if ext.outFormat == _lib.OUT_RGBA16F else torch.float32
    if ext.tileSize > 0:
        n = tiles_for_rank(params_w, params_h, ext.tileSize, ext.tileRank, ext.tileWorld)
        shape = (n, ext.tileSize, ext.tileSize, 4)
    else:
        shape = (params_h, params_w, 4)
    if out is None:
        return torch.empty(shape, dtype=dt, device=dev), params_w
    if out.dtype != dt or not out.is_cuda or out.stride(-1) != 1:
        raise TypeError(f"out: expected a device {dt} tensor with unit channel stride")
    if ext.tileSize > 0:
        if tuple(out.shape) != shape or not out.is_contiguous():
            raise ValueError(f"out: expected contiguous {shape}")
        return out, params_w
    if out.shape[0] < params_h or out.shape[1] < params_w or out.shape[2] != 4 or out.stride(1) != 4:
        raise ValueError(f"out: expected at least ({params_h},{params_w},4) with pixel stride 4")
    return out, out.stride(0) // 4


def _bind_brats(params, intensities, labels, preds, ext, dev, pred_stream: bool = False):
    """gParams dict + bound grids -> (MrirtBratsParams, MrirtRenderExt, device tensors), with the
    layout / size checks every K1 entry point shares."""
    P = brats_params(params)
    e = dict(ext or {})
    vgrids = [g for g in intensities if isinstance(g, Grid)]
    lgrids = [g for g in (labels, preds) if isinstance(g, Grid)]
    if vgrids and "layout" not in e:
        e["layout"] = vgrids[0].layout
    if lgrids and "labelLayout" not in e:
        e["labelLayout"] = lgrids[0].layout
    E = render_ext(e)
    names = {v: k for k, v in (("linear", 0), ("brick", 1), ("vg", 2), ("quad", 3), ("vga", 4), ("labcell", 5))}
    lay, lab_lay = names[E.layout], names[E.labelLayout]
    for g, want in [(g, lay) for g in vgrids] + [(g, lab_lay) for g in lgrids]:
        if g.layout != want:
            raise ValueError(f"bound grid is {g.layout!r} but the render call says {want!r}")
        if tuple(g.dims) != tuple(int(v) for v in P.dims):
            raise ValueError(f"grid dims {g.dims} != gParams.dims {tuple(P.dims)}")
    dims = tuple(int(v) for v in P.dims)
    nvox = dims[0] * dims[1] * dims[2]
    need = {"linear": nvox, "brick": brick_elems(dims), "vg": 4 * vec4_elems(dims), "quad": 4 * vec4_elems(dims),
            "vga": 4 * vga_elems(dims), "labcell": 2 * vec4_elems(dims)}
    if lab_lay == "labcell" and lay != "quad":
        raise ValueError("label cells (upload_label_cells) go with 'quad' intensity grids")
    vols = []
    for m in range(4):
        v = intensities[m] if m < len(intensities) else None
        t = _as_device_tensor(v, torch.float32, dev, f"gIntensity{m}")
        if P.volEnabled[m] != 0:
            if t is None or t.numel() < need[lay]:
                raise ValueError(f"gIntensity{m} is enabled but holds {0 if t is None else t.numel()} < {need[lay]} elements")
        vols.append(t)
    lab = _as_device_tensor(labels, torch.int32, dev, "gLabels")
    prd = _as_device_tensor(preds, torch.int32, dev, "gPreds")
    if P.showSeg != 0 and (lab is None or lab.numel() < need[lab_lay]):
        raise ValueError("showSeg is set but gLabels is missing or too small")
    if lab_lay == "labcell":
        if pred_stream:
            raise ValueError("label cells carry the prediction grid: not with a class stream")
        if P.showPred != 0 and (lab is None or lab.numel() < need[lab_lay]):
            raise ValueError("showPred is set but the label-cell grid is missing or too small")
        prd = None
    elif P.showPred != 0 and not pred_stream and (prd is None or prd.numel() < need[lab_lay]):
        raise ValueError("showPred is set but gPreds is missing or too small")
    return P, E, vols, lab, prd


_last_skip_mask: Optional[torch.Tensor] = None
_SKIP_MAPS: "collections.OrderedDict[tuple, torch.Tensor]" = collections.OrderedDict()
_SKIP_MAPS_MAX = 8
skip_map_builds = 0              # how many launches built a map (tests: a repeated frame must not)


def _bind_skip(P, E, intensities, labels, preds, dev, stream):
    """MrirtSkip for one call: the macro summaries of the bound Grid objects + the mask / empty-radius map scratch.

    The map depends on the grids' summaries and on (dims, ww, wl, gamma, volEnabled, volWeight, showSeg, showPred, math) —
    not on the camera — so a viewer's frames reuse it: scratches are cached per (those values, the bound summary
    tensors, device, stream) and a hit sets ``mapReady`` (the library then skips the four pre-pass launches, ~30 us of
    a 160 us viewer frame).  The stream is part of the key because the map was written by a launch on that stream and
    nothing else orders a reader on another stream after it; the summary tensors are held by the entry, so a key can never
    name freed-and-reused memory."""
    global _last_skip_mask, skip_map_builds
    S = _lib.Skip()
    keep = []
    for m in range(4):
        g = intensities[m] if m < len(intensities) else None
        if P.volEnabled[m] != 0:
            if not isinstance(g, Grid) or g.macro is None or g.macro.dtype != torch.float32:
                raise ValueError(f"skip=True: gIntensity{m} must be a Grid made by upload_grid (it carries the macro-cell bounds)")
            S.macroUb[m] = g.macro.data_ptr()
            keep.append(g.macro)
    cells = isinstance(labels, Grid) and labels.layout == "labcell"      # one grid carries both summaries
    for name, g, flag, which in (("macroSeg", labels, P.showSeg, "macro"), ("macroPred", labels if cells else preds, P.showPred, "macro2" if cells else "macro")):
        if flag != 0:
            m = getattr(g, which, None) if isinstance(g, Grid) else None
            if m is None or m.dtype != torch.int32:
                raise ValueError("skip=True: a shown label grid must be a Grid made by upload_grid / upload_label_cells")
            setattr(S, name, m.data_ptr())
            keep.append(m)
    d = (C.c_uint32 * 3)(*[int(v) for v in P.dims])
    words = int(_lib.lib().mrirt_skip_mask_words(d))
    sid = stream if stream is not None else torch.cuda.current_stream()
    sid = sid.cuda_stream if hasattr(sid, "cuda_stream") else int(sid)
    key = (tuple(int(v) for v in P.dims), float(P.ww), float(P.wl), float(P.gamma), tuple(int(v) for v in P.volEnabled),
           tuple(float(v) for v in P.volWeight), int(P.showSeg) != 0, int(P.showPred) != 0, int(E.math),
           tuple(t.data_ptr() for t in keep), dev.index, sid)
    hit = _SKIP_MAPS.get(key)
    if hit is not None:
        _SKIP_MAPS.move_to_end(key)
        mask, S.mapReady = hit[0], 1
        if hit[2][0] is None:
            # second frame with this map: how much of the volume IS empty?  (One read-back per map, i.e. per change of window /
            # weights / overlays — never per camera move.)  Where almost nothing can be skipped the skipping march only costs
            # (a dense synthetic head: 0.161 vs 0.137 ms per viewer frame), so such maps switch it off for their lifetime.
            cells = int(_lib.lib().mrirt_macro_cells(d))
            bits = mask[: ((cells + 63) // 64) * 2].view(torch.uint8)
            hit[2][0] = float(_POPCOUNT.to(dev)[bits.long()].sum().item()) / max(cells, 1)
        if hit[2][0] < SKIP_MIN_EMPTY_FRACTION:
            return None, None
    else:
        mask = torch.empty(words, dtype=torch.int32, device=dev)
        _SKIP_MAPS[key] = (mask, list(keep), [None])    # holds the summaries: their addresses cannot be recycled under the key
        while len(_SKIP_MAPS) > _SKIP_MAPS_MAX:
            _SKIP_MAPS.popitem(last=False)
        skip_map_builds += 1
    S.mask, S.maskWords = mask.data_ptr(), words
    keep.append(mask)
    _last_skip_mask = mask          # inspection hook (tests read the fraction of skippable cells)
    return S, keep


SKIP_MIN_EMPTY_FRACTION = 0.10   # of the 8^3 macro cells; below it skip=True renders with the plain kernels (same bits)
_POPCOUNT = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int32)


def render_brats(params: Mapping[str, Any], intensities: Sequence[Optional[Union[ArrayLike, Grid]]],
                 labels: Optional[Union[ArrayLike, Grid]] = None, preds: Optional[Union[ArrayLike, Grid]] = None,
                 out: Optional[torch.Tensor] = None, ext: Optional[Mapping[str, Any]] = None,
                 stats: bool = False, stream=None, skip: bool = False):
    """K1 — drop-in for ``kernel.dispatch`` of ``brats_main`` (inr/viewer/brats_viewer.py:431-442).

    ``params`` is the reference's ``gParams`` dict; ``intensities`` are ``gIntensity0..3``,
    ``labels``/``preds`` are ``gLabels``/``gPreds``.  Returns the fp32 (H,W,4) frame (or the
    compact tile buffer when ``ext`` shards tiles).  With ``stats=True`` also returns
    ``{"live_samples", "shaded_samples"}`` counted on the device.

    ``skip=True`` turns on exact empty-space skipping (same bits, same counters; needs the bound grids to be
    ``Grid`` objects carrying their macro-cell summaries, i.e. made by ``upload_grid``): samples in 8^3 macro
    cells that cannot contribute under this call's window, weights and overlays composite nothing, and the packet
    crosses wide empty regions in leaps sized by an empty-radius map of those cells.
    """
    dev = _require_gpu()
    # every temporary of this call (uploads of host arrays, the output, the counters, the skip mask) is made on
    # the launch stream, so the kernels are ordered after their producers and the caching allocator ties the
    # memory to that stream
    with _on_stream(stream):
        P, E, vols, lab, prd = _bind_brats(params, intensities, labels, preds, ext, dev)
        o, pitch = _alloc_out(int(P.imageSize[0]), int(P.imageSize[1]), E, dev, out)
        vp = (C.c_void_p * 4)(*[C.c_void_p(t.data_ptr()) if t is not None else None for t in vols])
        st = torch.zeros(2, dtype=torch.int64, device=dev) if stats else None
        S = None
        if skip:
            S, keep = _bind_skip(P, E, intensities, labels, preds, dev, stream)
        if S is not None:
            rc = _lib.lib().mrirt_render_brats_skip(C.byref(P), C.byref(E), vp, _ptr(lab), _ptr(prd), C.byref(S), _ptr(o),
                                                    pitch, _ptr(st), _stream_ptr(stream))
            _lib.check(rc, "mrirt_render_brats_skip")
            del keep
        else:
            rc = _lib.lib().mrirt_render_brats_ex(C.byref(P), C.byref(E), vp, _ptr(lab), _ptr(prd), _ptr(o),
                                                  pitch, _ptr(st), _stream_ptr(stream))
            _lib.check(rc, "mrirt_render_brats_ex")
        if stats:
            s = st.cpu()
            return o, {"live_samples": int(s[0]), "shaded_samples": int(s[1])}
    return o


_VOX = {"u32x4": (_lib.VOX_U32X4, torch.int32), "u8": (_lib.VOX_U8, torch.uint8), "f32": (_lib.VOX_F32, torch.float32),
        "cell8": (_lib.VOX_CELL8, torch.int64)}


def build_cell8(volume: ArrayLike, dims: Sequence[int], mode: str = "u8", stream=None) -> torch.Tensor:
    """Load time: u8 voxels (``mode='u8'``) or the reference's one-uint32-per-voxel upload (``'u32x4'``,
    scripts/volumeRendering/app.py:149-153) -> the CELL8 grid ``render_volume_u8(..., mode='cell8')`` marches:
    every voxel carries the eight bytes of its trilinear cell, so a sample is ONE 8-byte gather (same frame)."""
    dev = _require_gpu()
    code, dt = _VOX[mode]
    if mode not in ("u8", "u32x4"):
        raise ValueError("build_cell8 takes u8 or u32x4 voxels")
    t = torch.as_tensor(volume)
    if mode == "u32x4" and t.dtype in (torch.uint32, torch.int64):
        t = t.to(torch.int64).to(torch.int32)
    if t.dtype != dt:
        raise TypeError(f"gVolumeU8 ({mode}): expected {dt}, got {t.dtype}")
    dims = tuple(int(v) for v in dims)
    n = dims[0] * dims[1] * dims[2]
    t = t.to(dev).contiguous().reshape(-1)
    if t.numel() < n:
        raise ValueError(f"gVolumeU8 holds {t.numel()} < {n} voxels")
    out = torch.empty(n, dtype=torch.int64, device=dev)
    d = (C.c_uint32 * 3)(*dims)
    _lib.check(_lib.lib().mrirt_build_cell8(_ptr(t), code, d, _ptr(out), _stream_ptr(stream)), "mrirt_build_cell8")
    return out


def render_volume_u8(params: Mapping[str, Any], volume: ArrayLike, mode: str = "u32x4",
                     out: Optional[torch.Tensor] = None, ext: Optional[Mapping[str, Any]] = None,
                     stats: bool = False, stream=None):
    """K2 — drop-in for ``kernel.dispatch`` of ``volume_cs`` (scripts/volumeRendering/app.py:350-358).
    ``volume`` is ``gVolumeU8``: the (N/4,4) uint32 array of app.py:153 (``mode='u32x4'``), real
    bytes (``'u8'``) or an fp32 grid (``'f32'``)."""
    dev = _require_gpu()
    P = volume_params(params)
    E = render_ext(ext)
    code, dt = _VOX[mode]
    t = torch.as_tensor(volume)
    if mode == "u32x4" and t.dtype in (torch.uint32, torch.int64):
        t = t.to(torch.int64).to(torch.int32)
    if t.dtype != dt:
        raise TypeError(f"gVolumeU8 ({mode}): expected {dt}, got {t.dtype}")
    with _on_stream(stream):                               # uploads and temporaries on the launch stream
        t = t.to(dev).contiguous().reshape(-1)
        nvox = int(P.volDim[0]) * int(P.volDim[1]) * int(P.volDim[2])
        if t.numel() < nvox:
            raise ValueError(f"gVolumeU8 holds {t.numel()} < {nvox} voxels")
        o, pitch = _alloc_out(int(P.imageSize[0]), int(P.imageSize[1]), E, dev, out)
        st = torch.zeros(2, dtype=torch.int64, device=dev) if stats else None
        rc = _lib.lib().mrirt_render_volume(C.byref(P), C.byref(E), _ptr(t), code, _ptr(o), pitch, _ptr(st),
                                            _stream_ptr(stream))
        _lib.check(rc, "mrirt_render_volume")
        if stats:
            return o, {"live_samples": int(st.cpu()[0])}
    return o


def render_sdf(params: Mapping[str, Any], eye, U, V, W, width: int, height: int,
               out: Optional[torch.Tensor] = None, stream=None) -> torch.Tensor:
    """K3 — drop-in for ``kernel.dispatch`` of ``raymarch_cs`` (scripts/raymarch/app.py:212-223);
    ``width``/``height`` are the render_texture's dimensions."""
    dev = _require_gpu()
    P = sdf_params(params, eye, U, V, W)
    if out is None:
        out = torch.empty((height, width, 4), dtype=torch.float32, device=dev)
    pitch = out.stride(0) // 4
    rc = _lib.lib().mrirt_render_sdf(C.byref(P), int(width), int(height), _ptr(out), pitch, _stream_ptr(stream))
    _lib.check(rc, "mrirt_render_sdf")
    return out


def detile(gathered: torch.Tensor, width: int, height: int, tile: int, world: int,
           out: Optional[torch.Tensor] = None, stream=None) -> torch.Tensor:
    """[world, max_local, ts, ts, 4] compact tiles -> (H, W, 4) frame (csrc/grid_ops.hip)."""
    half = gathered.dtype == torch.float16
    if out is None:
        out = torch.empty((height, width, 4), dtype=gathered.dtype, device=gathered.device)
    rc = _lib.lib().mrirt_detile(_ptr(gathered), _ptr(out), width, height, out.stride(0) // 4, tile, world,
                                 _lib.OUT_RGBA16F if half else _lib.OUT_RGBA32F, _stream_ptr(stream))
    _lib.check(rc, "mrirt_detile")
    return out
