"""slangpy-shaped front end: ``Device`` / ``Buffer`` / ``Texture`` / ``ComputeKernel.dispatch``.

The reference's only render interface is the third-party slangpy call
    kernel = device.create_compute_kernel(device.load_program("brats_rt.slang", ["brats_main"]))
    kernel.dispatch(thread_count=[W, H, 1], vars={...}, command_encoder=ce)
(inr/viewer/brats_viewer.py:85-86,431-442; scripts/volumeRendering/app.py:30-31,350-358;
scripts/raymarch/app.py:26-27,212-223).  This module offers the same call shape — binding BY
NAME to the Slang globals, ``gParams`` as a dict of cbuffer fields, caller-owned buffers and
output texture — backed by the gfx950 kernels, so viewer-style code swaps ``spy.Device`` for
``mrirt.shim.Device`` and nothing else.  Work is enqueued on the current HIP stream and not
synchronised (slangpy records into a command encoder and submits asynchronously, :449).
"""
from __future__ import annotations

import enum
import pathlib
from typing import Any, Dict, List, Mapping, Optional, Sequence, Union

import numpy as np
import torch

from . import render as _r

# entry point -> (kernel id, the Slang globals it binds)
_ENTRY_POINTS = {
    "brats_main": ("K1", {"gOutput", "gIntensity0", "gIntensity1", "gIntensity2", "gIntensity3",
                          "gLabels", "gPreds", "gParams"}),
    "volume_cs": ("K2", {"gOutput", "gParams", "gVolumeU8"}),
    "raymarch_cs": ("K3", {"render_texture", "gParams", "gEye", "gU", "gV", "gW"}),
}


class Format(enum.Enum):
    rgba16_float = "rgba16_float"
    rgba32_float = "rgba32_float"


class BufferUsage(enum.Flag):
    shader_resource = 1
    unordered_access = 2


TextureUsage = BufferUsage


class Buffer:
    """Device buffer (``device.create_buffer`` + ``copy_from_numpy``).  A bricked copy for the
    march kernels is derived lazily per grid shape and dropped when the contents change."""

    def __init__(self, device: "Device", element_count: int, struct_size: int = 4):
        self.device, self.element_count, self.struct_size = device, int(element_count), int(struct_size)
        self.tensor: Optional[torch.Tensor] = None
        self._bricked: Dict[tuple, _r.Grid] = {}
        self._version = 0                                  # bumped by every rewrite (derived copies shared by two buffers key on it)

    def copy_from_numpy(self, arr: np.ndarray) -> None:
        a = np.ascontiguousarray(arr)
        if a.nbytes > self.element_count * self.struct_size:
            raise ValueError(f"copy_from_numpy: {a.nbytes} bytes into a {self.element_count * self.struct_size}-byte buffer")
        if a.dtype == np.uint32:
            a = a.view(np.int32)
        self.tensor = torch.from_numpy(a.reshape(-1)).to(self.device.torch_device)
        self._bricked.clear()
        self._version += 1

    @classmethod
    def from_numpy(cls, device: "Device", arr: np.ndarray) -> "Buffer":
        b = cls(device, arr.size, arr.dtype.itemsize)
        b.copy_from_numpy(arr)
        return b

    def to_numpy(self) -> np.ndarray:
        return self.tensor.cpu().numpy()

    def grid(self, dims: Sequence[int], layout: str) -> Union[torch.Tensor, _r.Grid]:
        if self.tensor is None:
            raise ValueError("buffer was never filled (copy_from_numpy)")
        dims = tuple(int(d) for d in dims)
        if layout == "linear" or self.tensor.numel() < dims[0] * dims[1] * dims[2]:
            return _r.Grid(self.tensor, dims, "linear")
        key = (dims, layout)
        if key not in self._bricked:
            self._bricked[key] = _r.upload_grid(self.tensor[:dims[0] * dims[1] * dims[2]], dims, layout)
        return self._bricked[key]


class Texture:
    def __init__(self, device: "Device", format: Format, width: int, height: int):
        self.format, self.width, self.height = Format(format), int(width), int(height)
        dt = torch.float16 if self.format is Format.rgba16_float else torch.float32
        self.tensor = torch.zeros((self.height, self.width, 4), dtype=dt, device=device.torch_device)

    def to_numpy(self) -> np.ndarray:
        return self.tensor.cpu().numpy()


class CommandEncoder:
    """Dispatches are enqueued immediately on the current stream (already asynchronous), so the
    encoder only preserves the reference's call shape: dispatch -> blit -> finish -> submit."""

    def blit(self, dst: Texture, src: Texture) -> None:
        dst.tensor.copy_(src.tensor.to(dst.tensor.dtype), non_blocking=True)

    def finish(self) -> "CommandEncoder":
        return self


class Program:
    def __init__(self, path: str, entry_points: Sequence[str]):
        self.path, self.entry_points = str(path), list(entry_points)


class ComputeKernel:
    def __init__(self, device: "Device", program: Program):
        if len(program.entry_points) != 1:
            raise ValueError("one compute entry point per kernel")
        self.device, self.entry = device, program.entry_points[0]
        self.kind, self._globals = _ENTRY_POINTS[self.entry]

    def dispatch(self, thread_count: Sequence[int], vars: Mapping[str, Any],
                 command_encoder: Optional[CommandEncoder] = None, ext: Optional[Mapping[str, Any]] = None) -> None:
        unknown = set(vars) - self._globals
        if unknown:
            raise KeyError(f"{self.entry}: no such shader globals {sorted(unknown)}")
        missing = self._globals - set(vars)
        if missing:
            raise KeyError(f"{self.entry}: unbound shader globals {sorted(missing)}")
        tc = [int(v) for v in thread_count]
        if self.kind == "K1":
            self._brats(tc, vars, ext)
        elif self.kind == "K2":
            self._volume(tc, vars, ext)
        else:
            self._sdf(tc, vars)

    # -- helpers -------------------------------------------------------------------------
    def _out(self, tex: Texture, tc: List[int], ext: Optional[Mapping[str, Any]]) -> Dict[str, Any]:
        if not isinstance(tex, Texture):
            raise TypeError("gOutput must be a Texture created by device.create_texture")
        if tc[0] != tex.width or tc[1] != tex.height:
            raise ValueError(f"thread_count {tc[:2]} must equal the output texture size {[tex.width, tex.height]}")
        e = dict(self.device.default_ext)
        e.update(ext or {})
        e["outFormat"] = "rgba16f" if tex.format is Format.rgba16_float else "rgba32f"
        return e

    def _label_cells(self, seg: Optional[Buffer], pred: Optional[Buffer], dims) -> "_r.Grid":
        key = (dims, id(seg), seg._version if seg else 0, id(pred), pred._version if pred else 0)
        cache = self.device._label_cells
        hit = cache.get(key)
        if hit is None:
            n = dims[0] * dims[1] * dims[2]
            g = _r.upload_label_cells(seg.tensor[:n] if seg else None, pred.tensor[:n] if pred else None, dims)
            cache.clear()                                   # one pair at a time (the viewer has one); the entry keeps its buffers alive
            hit = cache[key] = (g, seg, pred)
        return hit[0]

    def _mod4(self, bufs: Sequence[Optional[Buffer]], dims) -> "_r.Grid":
        """The enabled intensity buffers as ONE float4 grid (render.upload_mod4; a disabled modality is stored as zeros),
        cached until one of them is rewritten."""
        key = (dims,) + tuple((id(b), b._version) if b is not None else (0, 0) for b in bufs)
        cache = self.device._mod4
        hit = cache.get(key)
        if hit is None:
            n = dims[0] * dims[1] * dims[2]
            g = _r.upload_mod4([b.tensor[:n] if b is not None else None for b in bufs], dims)
            while len(cache) >= 3:                          # a few enabled-sets of one case (the viewer's check boxes toggle between
                cache.pop(next(iter(cache)))                # them); an entry keeps its buffers alive
            hit = cache[key] = (g, tuple(bufs))
        else:
            cache[key] = cache.pop(key)                     # most recently used last
        return hit[0]

    def _buf(self, b, dims, layout):
        if isinstance(b, Buffer):
            return b.grid(dims, layout)
        return b          # numpy array or torch tensor in the linear layout

    def _brats(self, tc, vars, ext):
        p = vars["gParams"]
        tex = vars["gOutput"]
        e = self._out(tex, tc, ext)
        if (int(p["imageSize"][0]), int(p["imageSize"][1])) != (tex.width, tex.height):
            raise ValueError("gParams.imageSize must equal the output texture size")
        dims = [int(v) for v in p["dims"]]
        layout = e.get("layout", "linear")
        raw = [vars[f"gIntensity{m}"] for m in range(4)] + [vars["gLabels"], vars["gPreds"]]
        if not all(isinstance(b, Buffer) for b in raw):
            layout = "linear"
        # "auto": float4 voxels — value+gradient in axis-flat bricks (three copies, "vga") when shading, xy-quads in
        # 2x2x2 bricks otherwise; label grids in 4x4x2 bricks.  Derived copies are cached on the Buffer until it is
        # rewritten.
        vlay = ("vga" if int(e.get("shadeMode", 0)) else "quad") if layout == "auto" else layout
        llay = "linear" if layout == "linear" else "brick"
        # dummy 1-element buffers stand in for disabled inputs (brats_viewer.py:247-248,437-438)
        en = [int(v) != 0 for v in p["volEnabled"]] + [int(p["showSeg"]) != 0, int(p["showPred"]) != 0]
        nvox = dims[0] * dims[1] * dims[2]
        # "auto", unshaded, two or more modalities: ONE float4 grid of the enabled modalities (MRIRT_LAYOUT_MOD4) instead of a
        # quad grid each — the same gathers per sample out of a quarter of the memory (the viewer's four-modality frame)
        if layout == "auto" and vlay == "quad" and sum(en[:4]) >= 2 and all((not on) or (b.tensor is not None and b.tensor.numel() >= nvox)
                                                                            for b, on in zip(raw[:4], en[:4])):
            vlay = "mod4"
        lays = [vlay] * 4 + [llay] * 2
        if vlay == "mod4":
            m4 = self._mod4([b if on else None for b, on in zip(raw[:4], en[:4])], tuple(dims))
            bound = [m4] * 4 + [self._buf(b, dims, lay) if on else None for b, on, lay in zip(raw[4:], en[4:], lays[4:])]
        else:
            bound = [self._buf(b, dims, lay) if on else None for b, on, lay in zip(raw, en, lays)]
        # QUAD voxels + overlays: both label buffers as ONE cell-packed grid (render.upload_label_cells): a sample then takes one
        # 8-byte gather at its cell's own offset instead of two nearest-voxel gathers (same bits; the viewer's frame -10 %)
        if vlay in ("quad", "mod4") and (en[4] or en[5]) and all((not on) or (b.tensor is not None and b.tensor.numel() >= nvox)
                                                       for b, on in zip(raw[4:], en[4:])):
            bound[4], bound[5] = self._label_cells(raw[4] if en[4] else None, raw[5] if en[5] else None, tuple(dims)), None
            llay = "labcell"
        e["layout"], e["labelLayout"] = vlay, llay
        # exact empty-space skipping rides on the macro-cell summaries upload_grid attached to the cached grids
        skip = (self.device.skip_empty and vlay in ("vg", "vga", "quad", "mod4")
                and all(b is None or (isinstance(b, _r.Grid) and (b.macro is not None or b.macros is not None)) for b in bound))
        _r.render_brats(p, bound[:4], bound[4], bound[5], out=tex.tensor, ext=e, skip=skip)

    def _volume(self, tc, vars, ext):
        p, tex = vars["gParams"], vars["gOutput"]
        e = self._out(tex, tc, ext)
        e.pop("layout", None)
        v = vars["gVolumeU8"]
        if isinstance(v, Buffer) and self.device.default_ext.get("layout") != "linear":
            # 'auto': march the CELL8 copy of the reference's u32-per-voxel upload (one gather per sample,
            # same frame); derived once per buffer contents, like K1's bricked copies
            dims = tuple(int(d) for d in p["volDim"])
            key = (dims, "cell8")
            if key not in v._bricked:
                v._bricked[key] = _r.build_cell8(v.tensor, dims, mode="u32x4")
            _r.render_volume_u8(p, v._bricked[key], mode="cell8", out=tex.tensor, ext=e)
            return
        t = v.tensor if isinstance(v, Buffer) else v
        _r.render_volume_u8(p, t, mode="u32x4", out=tex.tensor, ext=e)

    def _sdf(self, tc, vars):
        tex = vars["render_texture"]
        if tc[0] != tex.width or tc[1] != tex.height:
            raise ValueError("thread_count must equal the render_texture size")
        if tex.format is Format.rgba32_float:
            _r.render_sdf(vars["gParams"], vars["gEye"], vars["gU"], vars["gV"], vars["gW"], tex.width, tex.height, out=tex.tensor)
        else:
            tex.tensor.copy_(_r.render_sdf(vars["gParams"], vars["gEye"], vars["gU"], vars["gV"], vars["gW"], tex.width, tex.height))


class Device:
    """Stands where ``spy.Device(...)`` stands.  ``layout='auto'`` (default) lets K1 march a
    float4-bricked copy of each bound Buffer (see ComputeKernel._brats; 'linear' binds the buffers
    as uploaded, like the reference); ``math`` selects the strict or fast arithmetic flavour;
    ``skip_empty`` (default on; it changes no bit of the frame) lets the march skip 8^3 macro cells that
    cannot contribute under the frame's window / overlays — the air around a skull-stripped scan."""

    def __init__(self, enable_debug_layers: bool = False, compiler_options: Optional[dict] = None,
                 layout: str = "auto", math: str = "strict", skip_empty: bool = True):
        self.torch_device = _r._require_gpu()
        self.default_ext = {"layout": layout, "math": math}
        self.skip_empty = bool(skip_empty)
        self._label_cells: Dict[tuple, tuple] = {}          # the cell-packed copy of the bound (gLabels, gPreds) pair
        self._mod4: Dict[tuple, tuple] = {}                 # the interleaved copy of the bound intensity buffers

    def load_program(self, path: Union[str, pathlib.Path], entry_points: Sequence[str]) -> Program:
        for ep in entry_points:
            if ep not in _ENTRY_POINTS:
                raise RuntimeError(f"load_program({path}): no gfx950 kernel for entry point '{ep}' "
                                   f"(have {sorted(_ENTRY_POINTS)})")
        return Program(str(path), entry_points)

    def create_compute_kernel(self, program: Program) -> ComputeKernel:
        return ComputeKernel(self, program)

    def create_buffer(self, element_count: int, struct_size: int = 4, element_size: Optional[int] = None,
                      usage=None) -> Buffer:
        return Buffer(self, element_count, element_size if element_size is not None else struct_size)

    def create_texture(self, format: Format, width: int, height: int, usage=None) -> Texture:
        return Texture(self, format, width, height)

    def create_command_encoder(self) -> CommandEncoder:
        return CommandEncoder()

    def submit_command_buffer(self, cb) -> None:
        return None

    def wait(self) -> None:
        torch.cuda.synchronize(self.torch_device)


class KernelShim:
    """``KernelShim("brats_main").dispatch(thread_count, vars, command_encoder=None)`` — the
    one-object form named in SURVEY.md section 8(b)."""

    def __init__(self, entry_point: str, **device_kwargs):
        self.device = Device(**device_kwargs)
        self.kernel = self.device.create_compute_kernel(self.device.load_program(entry_point + ".slang", [entry_point]))

    def dispatch(self, thread_count, vars, command_encoder=None, ext=None):
        return self.kernel.dispatch(thread_count, vars, command_encoder, ext)
