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


TRACE_HOOK = None      # diagnostics: a callable(str) that receives the device addresses and sizes of every render_brats call's
                       # tensors (tests/conftest.py sets it for fault-attribution runs); None in normal use


def _require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("mrirt: no HIP device visible — the ray-marcher has no CPU fallback "
                           "(the CPU restatement under oracle/ is test infrastructure only)")
    return torch.device("cuda", torch.cuda.current_device())


def _stream_ptr(stream) -> C.c_void_p:
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream if hasattr(s, "cuda_stream") else int(s))


def _as_stream(stream) -> "torch.cuda.Stream":
    """The launch stream as a torch stream object: None -> torch's current stream; a raw hipStream_t (integer) is adopted
    as a ``torch.cuda.ExternalStream`` (torch does not own it), so that temporaries can be allocated on it too."""
    if stream is None:
        return torch.cuda.current_stream()
    if isinstance(stream, torch.cuda.Stream):
        return stream
    return torch.cuda.ExternalStream(int(stream))


def _on_stream(stream):
    """Context that makes `stream` torch's current stream: every temporary of a call (uploads, the output, counters, the
    skipping scratch) is then allocated ON the launch stream, which is what ties its lifetime to the work that uses it —
    the caching allocator hands a freed block to a later allocation of the SAME stream only, i.e. behind the kernels
    that still read it.  A no-op for the default (None)."""
    if stream is None:
        return contextlib.nullcontext()
    return torch.cuda.stream(_as_stream(stream))


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(t.data_ptr() if t is not None else None)


@dataclass
class Grid:
    """A device-resident voxel grid: the analogue of the reference's StructuredBuffer created by
    ``device.create_buffer`` + ``copy_from_numpy`` (inr/viewer/brats_viewer.py:219-230)."""
    data: torch.Tensor            # 1-D, device
    dims: Tuple[int, int, int]    # (X, Y, Z), x fastest in the linear layout
    layout: str = "linear"        # "linear" | "brick" | "vg" | "quad" | "vga" | "mod4" | "labcell"
    macro: Optional[torch.Tensor] = None   # per 8^3 macro cell: fp32 upper bound (intensities) / any-label word
                                           # (label grids) — what exact empty-space skipping tests (skip=True)
    macro2: Optional[torch.Tensor] = None  # "labcell" grids: the prediction grid's summary (macro = the ground truth's)
    macros: Optional[Tuple[Optional[torch.Tensor], ...]] = None   # "mod4" grids: the four modalities' summaries (or None each)

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


def upload_mod4(modalities: Sequence[Optional[ArrayLike]], dims: Sequence[int], stream=None, macro: bool = True) -> Grid:
    """The four modalities of one case (linear fp32 grids, gIntensity0..3) as ONE "mod4" grid: float4 (m0, m1, m2, m3) per voxel
    in the "vg" grid's element order (include/mrirt.h, MRIRT_LAYOUT_MOD4).  An unshaded sample of a four-modality frame is the
    same eight 16-byte gathers as from four "quad" grids, out of a quarter of the memory (one set of cache lines per sample
    instead of four: config 2 0.40 -> 0.24 ms) — same bits.  Bind it four times (``[g] * 4``) to ``render_brats`` or pass it as
    ``intensities`` to ``render_brats_inr``; which modalities are drawn stays ``gParams.volEnabled``.  A modality given as
    ``None`` (one the frame never enables: the viewer binds a dummy buffer for it) is stored as zeros.  ``macro`` also builds
    the per-modality 8^3 summaries that ``skip=True`` needs."""
    dev = _require_gpu()
    dims = tuple(int(v) for v in dims)
    if len(modalities) != 4:
        raise ValueError("upload_mod4 takes the four modalities of a case")
    with _on_stream(stream):                             # the linear copies are temporaries of the launch stream
        ts, macros, zeros = [], [], None
        for m, v in enumerate(modalities):
            if v is None:
                if zeros is None:
                    zeros = torch.zeros(dims[0] * dims[1] * dims[2], dtype=torch.float32, device=dev)
                ts.append(zeros); macros.append(None)
                continue
            t = torch.as_tensor(v).reshape(-1).to(torch.float32).to(dev).contiguous()
            if t.numel() != dims[0] * dims[1] * dims[2]:
                raise ValueError(f"modality {m} has {t.numel()} voxels, dims {dims} need {dims[0] * dims[1] * dims[2]}")
            ts.append(t)
            macros.append(_build_macro(t, dims, stream) if macro else None)
        out = torch.empty(4 * vec4_elems(dims), dtype=torch.float32, device=dev)
        ptrs = (C.c_void_p * 4)(*[C.c_void_p(t.data_ptr()) for t in ts])
        _lib.check(_lib.lib().mrirt_build_mod4_grid(ptrs, _ptr(out), (C.c_uint32 * 3)(*dims), _stream_ptr(stream)), "mrirt_build_mod4_grid")
    return Grid(out, dims, "mod4", None, None, tuple(macros))


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
    names = {v: k for k, v in (("linear", 0), ("brick", 1), ("vg", 2), ("quad", 3), ("vga", 4), ("labcell", 5), ("mod4", 6))}
    lay, lab_lay = names[E.layout], names[E.labelLayout]
    for g, want in [(g, lay) for g in vgrids] + [(g, lab_lay) for g in lgrids]:
        if g.layout != want:
            raise ValueError(f"bound grid is {g.layout!r} but the render call says {want!r}")
        if tuple(g.dims) != tuple(int(v) for v in P.dims):
            raise ValueError(f"grid dims {g.dims} != gParams.dims {tuple(P.dims)}")
    dims = tuple(int(v) for v in P.dims)
    nvox = dims[0] * dims[1] * dims[2]
    need = {"linear": nvox, "brick": brick_elems(dims), "vg": 4 * vec4_elems(dims), "quad": 4 * vec4_elems(dims),
            "vga": 4 * vga_elems(dims), "labcell": 2 * vec4_elems(dims), "mod4": 4 * vec4_elems(dims)}
    if lab_lay == "labcell" and lay not in ("quad", "mod4"):
        raise ValueError("label cells (upload_label_cells) go with 'quad' or 'mod4' intensity grids")
    if lay == "mod4" and len(intensities) >= 1:          # ONE grid carries all four modalities: [g], [g] * 4 or (g, None, None, None)
        intensities = [intensities[0]] * 4
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


KERNEL_FAMILIES = {0: "none", 1: "generic", 2: "pipelined", 3: "rolling", 4: "slab", 5: "ring"}


def kernel_family(params: Mapping[str, Any], ext: Optional[Mapping[str, Any]] = None, skip: bool = False) -> Dict[str, Any]:
    """Which march kernel the library takes for a K1 launch with these parameters (``mrirt_brats_kernel_family``: a host-only
    query, no GPU needed, nothing is launched; the grids are stood in for by placeholder addresses — only ``gParams`` and
    ``ext`` decide).  ``{"family": "generic" | "pipelined" | "rolling" | "slab" | "ring" | "none", "skipping": bool,
    "label_cells": bool}``.  The generic kernel is the fall-back with 64-bit offsets: VG / QUAD grids >= 4 GiB, label grids
    >= 2^30 elements, LINEAR grids >= 2^30 voxels, BRICK grids, shaded LINEAR."""
    P = brats_params(params)
    E = render_ext(dict(ext or {}))
    fake = C.c_void_p(0x1000)
    vp = (C.c_void_p * 4)(*[fake if (P.volEnabled[m] != 0 or int(E.layout) == _lib.LAYOUT_MOD4) else None for m in range(4)])
    S = None
    if skip:
        S = _lib.Skip()
        for m in range(4):
            S.macroUb[m] = 0x1000 if P.volEnabled[m] != 0 else None
        S.macroSeg = S.macroPred = S.mask = 0x1000
        S.maskWords = 0xFFFFFFFF
    rc = int(_lib.lib().mrirt_brats_kernel_family(C.byref(P), C.byref(E), vp, fake, fake, C.byref(S) if S is not None else None))
    if rc < 0:
        _lib.check(rc, "mrirt_brats_kernel_family")
    return {"family": KERNEL_FAMILIES[rc & 15], "skipping": bool(rc & _lib.KERNEL_SKIPPING), "label_cells": bool(rc & _lib.KERNEL_LABEL_CELLS)}


_last_skip_mask: Optional[torch.Tensor] = None
skip_map_builds = 0              # how many launches built a map (tests: a repeated frame must not)
SKIP_MIN_EMPTY_FRACTION = 0.10   # of the 8^3 macro cells; below it skip=True renders with the plain kernels (same bits)
_POPCOUNT = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int32)


class _SkipMap:
    """One cached empty-radius map: the scratch tensor plus everything whose lifetime or identity the cache key rests on.

    Lifetimes (VERDICT r3 #1a).  The key names the macro summaries by ``id()``; the entry HOLDS those tensors, so an id —
    and the device address behind it — cannot be recycled while the key exists.  ``mask`` is allocated with the launch
    stream current (``_on_stream``), so when an entry is evicted (or a call's temporaries go out of scope) while kernels
    that read them are still queued, the caching allocator can hand the block only to a later allocation of the same
    stream, which is ordered behind those kernels; nothing here is freed on the strength of the host's position in the
    code.  ``built`` turns true only after the library reported that this launch marches with a map
    (``mrirt_brats_skip_applicable``) and the launch that builds it was enqueued without error: a scratch that was never
    written is never passed with ``mapReady``."""
    __slots__ = ("mask", "held", "stream", "empty_fraction", "built")

    def __init__(self, mask, held, stream):
        self.mask, self.held, self.stream = mask, tuple(held), stream
        self.empty_fraction: Optional[float] = None
        self.built = False


_SKIP_MAPS: "collections.OrderedDict[tuple, _SkipMap]" = collections.OrderedDict()
_SKIP_MAPS_MAX = 8


def _bind_skip(P, E, vp, lab, prd, intensities, labels, preds, dev, stream):
    """MrirtSkip for one call, or (None, None, None) when this launch has no use for a map (it then goes out as the plain
    render call and no scratch exists).  Returns (S, entry, hold): ``hold`` keeps every tensor S points at alive until the
    caller drops it — after the launch has been enqueued, at the end of ``render_brats``.

    The map depends on the grids' summaries and on (dims, ww, wl, gamma, volEnabled, volWeight, showSeg, showPred, math) —
    not on the camera, and not on the layout / shading / kernel variant, which only decide WHETHER a launch marches with a
    map (asked of the library per call, below) — so a viewer's frames reuse it: one entry per (those values, the bound
    summary tensors, device, stream); a hit on a BUILT entry sets ``mapReady`` (the library then skips the four pre-pass
    launches, ~30 us of a 160 us viewer frame).  The stream is part of the key because the map was written by a
    launch on that stream and nothing else orders a reader on another stream after it."""
    global _last_skip_mask, skip_map_builds
    lib = _lib.lib()
    S = _lib.Skip()
    hold = []
    for m in range(4):
        g = intensities[m] if m < len(intensities) else None
        if P.volEnabled[m] != 0:
            mac = (g.macros[m] if g.macros is not None else None) if isinstance(g, Grid) and g.layout == "mod4" else getattr(g, "macro", None)
            if not isinstance(g, Grid) or mac is None or mac.dtype != torch.float32:
                raise ValueError(f"skip=True: gIntensity{m} must be a Grid made by upload_grid / upload_mod4 (it carries the macro-cell bounds)")
            S.macroUb[m] = mac.data_ptr()
            hold.append(mac)
    cells_grid = isinstance(labels, Grid) and labels.layout == "labcell"      # one grid carries both summaries
    for name, g, flag, which in (("macroSeg", labels, P.showSeg, "macro"),
                                 ("macroPred", labels if cells_grid else preds, P.showPred, "macro2" if cells_grid else "macro")):
        if flag != 0:
            m = getattr(g, which, None) if isinstance(g, Grid) else None
            if m is None or m.dtype != torch.int32:
                raise ValueError("skip=True: a shown label grid must be a Grid made by upload_grid / upload_label_cells")
            setattr(S, name, m.data_ptr())
            hold.append(m)
    # does the library march this launch with a map at all?  (host-only query; < 0: the render call's own error)
    rc = int(lib.mrirt_brats_skip_applicable(C.byref(P), C.byref(E), vp, _ptr(lab), _ptr(prd), C.byref(S)))
    if rc < 0:
        _lib.check(rc, "mrirt_brats_skip_applicable")
    if rc == 0:
        return None, None, None
    d = (C.c_uint32 * 3)(*[int(v) for v in P.dims])
    words = int(lib.mrirt_skip_mask_words(d))
    ts = _as_stream(stream)
    key = (tuple(int(v) for v in P.dims), float(P.ww), float(P.wl), float(P.gamma), tuple(int(v) for v in P.volEnabled),
           tuple(float(v) for v in P.volWeight), int(P.showSeg) != 0, int(P.showPred) != 0, int(E.math),
           tuple(id(t) for t in hold), dev.index, int(ts.cuda_stream))
    entry = _SKIP_MAPS.get(key)
    if entry is not None and entry.built:
        _SKIP_MAPS.move_to_end(key)
        S.mapReady = 1
        if entry.empty_fraction is None:
            # second frame with this map: how much of the volume IS empty?  (One read-back per map, i.e. per change of window /
            # weights / overlays — never per camera move; it runs on the launch stream, behind the launch that built the map.)
            # Where almost nothing can be skipped the skipping march only costs (a dense synthetic head: 0.161 vs 0.137 ms per
            # viewer frame), so such maps switch it off for their lifetime.
            cells = int(lib.mrirt_macro_cells(d))
            bits = entry.mask[: ((cells + 63) // 64) * 2].view(torch.uint8)
            entry.empty_fraction = float(_POPCOUNT.to(dev)[bits.long()].sum().item()) / max(cells, 1)
        if entry.empty_fraction < SKIP_MIN_EMPTY_FRACTION:
            return None, None, None
    else:
        if entry is None:
            entry = _SKIP_MAPS[key] = _SkipMap(torch.empty(words, dtype=torch.int32, device=dev), hold, ts)
            while len(_SKIP_MAPS) > _SKIP_MAPS_MAX:
                _SKIP_MAPS.popitem(last=False)          # (its tensors: see _SkipMap — stream-ordered by the allocator)
        skip_map_builds += 1
    S.mask, S.maskWords = entry.mask.data_ptr(), words
    hold.append(entry.mask)
    _last_skip_mask = entry.mask    # inspection hook (tests read the fraction of skippable cells)
    return S, entry, hold


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
        if TRACE_HOOK is not None:
            TRACE_HOOK(("[render_brats] " + " ".join(f"{k}={t.data_ptr():#x}+{t.numel() * t.element_size():#x}" for k, t in
                         [(f"vol{m}", v) for m, v in enumerate(vols) if v is not None] + [("lab", lab), ("prd", prd), ("out", o), ("st", st)]
                         if t is not None) + f" skip={skip}\n"))
        S = entry = hold = None
        if skip:
            S, entry, hold = _bind_skip(P, E, vp, lab, prd, intensities, labels, preds, dev, stream)
        if S is not None:
            rc = _lib.lib().mrirt_render_brats_skip(C.byref(P), C.byref(E), vp, _ptr(lab), _ptr(prd), C.byref(S), _ptr(o),
                                                    pitch, _ptr(st), _stream_ptr(stream))
            _lib.check(rc, "mrirt_render_brats_skip")
            entry.built = True                      # the pre-pass is enqueued on the entry's stream: later frames may say mapReady
        else:
            rc = _lib.lib().mrirt_render_brats_ex(C.byref(P), C.byref(E), vp, _ptr(lab), _ptr(prd), _ptr(o),
                                                  pitch, _ptr(st), _stream_ptr(stream))
            _lib.check(rc, "mrirt_render_brats_ex")
        if stats:
            s = st.cpu()
            return o, {"live_samples": int(s[0]), "shaded_samples": int(s[1])}
        del hold                                    # (only here: every launch that reads these tensors has been enqueued)
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
           out: Optional[torch.Tensor] = None, stream=None, skew: int = 0) -> torch.Tensor:
    """[world, max_local, ts, ts, 4] compact tiles -> (H, W, 4) frame (csrc/grid_ops.hip); ``skew`` = the ``tileSkew`` the
    tiles were rendered with."""
    half = gathered.dtype == torch.float16
    if out is None:
        out = torch.empty((height, width, 4), dtype=gathered.dtype, device=gathered.device)
    rc = _lib.lib().mrirt_detile(_ptr(gathered), _ptr(out), width, height, out.stride(0) // 4, tile, world, int(skew),
                                 _lib.OUT_RGBA16F if half else _lib.OUT_RGBA32F, _stream_ptr(stream))
    _lib.check(rc, "mrirt_detile")
    return out
