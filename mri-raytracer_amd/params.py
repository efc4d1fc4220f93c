"""gParams dict  <->  C-ABI parameter structs.

The reference binds its constant buffers from plain dicts keyed by the Slang field names
(inr/viewer/brats_viewer.py:405-426, scripts/volumeRendering/app.py:334-345,
scripts/raymarch/app.py:202-209).  These helpers accept those dicts verbatim.
"""
from __future__ import annotations

from typing import Any, Dict, Mapping, Optional

import numpy as np

from . import _lib

# Build-defined extensions (all zero / absent = the reference's behaviour)
EXT_DEFAULTS: Dict[str, Any] = dict(
    cameraMode=0, orthoHalfHeight=1.1,
    shadeMode=0, ka=0.3, kd=0.6, ks=0.3, specPow2=5, gradEps=1e-6,
    ertThreshold=None,            # None -> the shader's hard-coded 0.01 (brats_rt.slang:117)
    math="strict",                # "strict" (bit-faithful) | "fast"
    outFormat="rgba32f",          # "rgba32f" | "rgba16f" (the reference's texture format)
    layout="linear",              # intensity grids: "linear" | "brick" | "vg" | "quad" | "vga" | "mod4" (unshaded frames)
    labelLayout="linear",         # labels / preds: "linear" | "brick" | "labcell" (both overlays per cell: upload_label_cells)
    tileSize=0, tileRank=0, tileWorld=0, tileSkew=0,
    kernelVariant=0,
)

_MATH = {"strict": _lib.MATH_STRICT, "fast": _lib.MATH_FAST}
_FMT = {"rgba32f": _lib.OUT_RGBA32F, "rgba16f": _lib.OUT_RGBA16F}
_LAYOUT = {"linear": _lib.LAYOUT_LINEAR, "brick": _lib.LAYOUT_BRICK, "vg": _lib.LAYOUT_VG, "quad": _lib.LAYOUT_QUAD,
           "vga": _lib.LAYOUT_VGA, "labcell": _lib.LAYOUT_LABCELL, "mod4": _lib.LAYOUT_MOD4}

_BRATS_REQUIRED = ("imageSize", "fovY", "eye", "U", "V", "W", "volMin", "voxelSize", "dims", "stepSize",
                   "nearT", "farT", "bgColor", "volEnabled", "volWeight", "ww", "wl", "intensityAlpha",
                   "gamma", "showSeg", "showPred", "lutColorAlpha")


def _set3(dst, src, name):
    a = np.asarray(src, dtype=np.float32).reshape(-1)
    if a.size < 3:
        raise TypeError(f"gParams.{name}: expected 3 floats, got {a.size}")
    for k in range(3):
        dst[k] = float(a[k])


def brats_params(p: Mapping[str, Any]) -> _lib.BratsParams:
    """dict (brats_viewer.py:405-426 field names) -> MrirtBratsParams."""
    missing = [k for k in _BRATS_REQUIRED if k not in p]
    if missing:
        # slangpy raises on unbound cbuffer fields too; keep the failure loud and named
        raise KeyError(f"gParams is missing fields: {missing}")
    s = _lib.BratsParams()
    s.imageSize[0], s.imageSize[1] = int(p["imageSize"][0]), int(p["imageSize"][1])
    s.fovY = float(np.float32(p["fovY"]))
    for k in ("eye", "U", "V", "W", "volMin", "voxelSize", "bgColor"):
        _set3(getattr(s, k), p[k], k)
    for k in range(3):
        s.dims[k] = int(p["dims"][k])
    for k in ("stepSize", "nearT", "farT", "ww", "wl", "intensityAlpha", "gamma"):
        setattr(s, k, float(np.float32(p[k])))
    s.gradBoost = float(p.get("gradBoost", 1.5))      # bound by the viewer, unread by the shader
    s.gradScale = float(p.get("gradScale", 1.0))
    for k in range(4):
        s.volEnabled[k] = int(p["volEnabled"][k])
        s.volWeight[k] = float(np.float32(p["volWeight"][k]))
    s.showSeg, s.showPred = int(p["showSeg"]), int(p["showPred"])
    lut = np.asarray(p["lutColorAlpha"], dtype=np.float32)
    if lut.shape != (8, 4):
        raise TypeError(f"gParams.lutColorAlpha: expected 8 float4 rows, got shape {lut.shape}")
    for i in range(8):
        for j in range(4):
            s.lutColorAlpha[i][j] = float(lut[i, j])
    return s


def render_ext(ext: Optional[Mapping[str, Any]] = None) -> _lib.RenderExt:
    e = dict(EXT_DEFAULTS)
    if ext:
        unknown = set(ext) - set(e)
        if unknown:
            raise KeyError(f"unknown render extension fields: {sorted(unknown)}")
        e.update(ext)
    s = _lib.RenderExt()
    s.cameraMode, s.orthoHalfHeight = int(e["cameraMode"]), float(np.float32(e["orthoHalfHeight"]))
    s.shadeMode, s.specPow2 = int(e["shadeMode"]), int(e["specPow2"])
    for k in ("ka", "kd", "ks", "gradEps"):
        setattr(s, k, float(np.float32(e[k])))
    if e["ertThreshold"] is not None:
        s.ertOverride, s.ertThreshold = 1, float(np.float32(e["ertThreshold"]))
    s.math, s.outFormat, s.layout = _MATH[e["math"]], _FMT[e["outFormat"]], _LAYOUT[e["layout"]]
    s.labelLayout = _LAYOUT[e["labelLayout"]]
    s.tileSize, s.tileRank, s.tileWorld = int(e["tileSize"]), int(e["tileRank"]), int(e["tileWorld"])
    s.tileSkew = int(e["tileSkew"])
    s.kernelVariant = int(e["kernelVariant"])
    return s


def volume_params(p: Mapping[str, Any]) -> _lib.VolumeParams:
    """dict (volumeRendering/app.py:334-345) -> MrirtVolumeParams."""
    s = _lib.VolumeParams()
    s.imageSize[0], s.imageSize[1] = int(p["imageSize"][0]), int(p["imageSize"][1])
    for k in ("fovY", "stepCount", "nearPlane", "farPlane"):
        setattr(s, k, float(np.float32(p[k])))
    for k in ("eye", "U", "V", "W"):
        _set3(getattr(s, k), p[k], k)
    for k in range(3):
        s.volDim[k] = int(p["volDim"][k])
    return s


def sdf_params(p: Mapping[str, Any], eye, U, V, W) -> _lib.SdfParams:
    """dict (raymarch/app.py:202-209) + gEye/gU/gV/gW -> MrirtSdfParams."""
    s = _lib.SdfParams()
    if "imageSize" in p:
        s.imageSize[0], s.imageSize[1] = int(p["imageSize"][0]), int(p["imageSize"][1])
    s.fovY = float(np.float32(p["fovY"]))
    s.maxSteps = int(p["maxSteps"])
    for k in ("maxDistance", "hitThreshold", "normalEps"):
        setattr(s, k, float(np.float32(p[k])))
    for name, v in (("gEye", eye), ("gU", U), ("gV", V), ("gW", W)):
        _set3(getattr(s, name), v, name)
    return s
