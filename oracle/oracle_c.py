"""ctypes front-end of the C oracle (oracle_c.c).  TEST INFRASTRUCTURE ONLY — see oracle_np.py.

Takes the same ``gParams``-style dicts as oracle_np so a test can run both restatements on
one input and demand bit-equality.
"""
from __future__ import annotations

import ctypes as C
import pathlib
import subprocess
from typing import Any, Dict, Optional, Sequence

import numpy as np

from .oracle_np import DEFAULT_EXT

_HERE = pathlib.Path(__file__).resolve().parent
_LIB: Optional[C.CDLL] = None


class BratsParams(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("fovY", C.c_float),
        ("eye", C.c_float * 3), ("U", C.c_float * 3), ("V", C.c_float * 3), ("W", C.c_float * 3),
        ("volMin", C.c_float * 3), ("voxelSize", C.c_float * 3), ("dims", C.c_uint32 * 3),
        ("stepSize", C.c_float), ("nearT", C.c_float), ("farT", C.c_float),
        ("bgColor", C.c_float * 3), ("volEnabled", C.c_uint32 * 4), ("volWeight", C.c_float * 4),
        ("ww", C.c_float), ("wl", C.c_float), ("intensityAlpha", C.c_float), ("gamma", C.c_float),
        ("showSeg", C.c_uint32), ("showPred", C.c_uint32), ("lut", (C.c_float * 4) * 8),
        ("cameraMode", C.c_uint32), ("orthoHalfHeight", C.c_float), ("shadeMode", C.c_uint32),
        ("ka", C.c_float), ("kd", C.c_float), ("ks", C.c_float), ("specPow2", C.c_uint32),
        ("gradEps", C.c_float), ("ertThreshold", C.c_float),
    ]


class VolumeParams(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("fovY", C.c_float), ("stepCount", C.c_float),
        ("nearPlane", C.c_float), ("farPlane", C.c_float),
        ("eye", C.c_float * 3), ("U", C.c_float * 3), ("V", C.c_float * 3), ("W", C.c_float * 3),
        ("volDim", C.c_uint32 * 3), ("mode", C.c_uint32), ("cameraMode", C.c_uint32),
        ("orthoHalfHeight", C.c_float),
    ]


class SdfParams(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("fovY", C.c_float), ("maxSteps", C.c_uint32),
        ("maxDistance", C.c_float), ("hitThreshold", C.c_float), ("normalEps", C.c_float),
        ("eye", C.c_float * 3), ("U", C.c_float * 3), ("V", C.c_float * 3), ("W", C.c_float * 3),
    ]


def build(force: bool = False) -> pathlib.Path:
    so = _HERE / "liboracle.so"
    src = _HERE / "oracle_c.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-B", "liboracle.so"], check=True, capture_output=True)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        import os
        so = _HERE / "liboracle.so"
        alt = os.environ.get("MRIRT_ORACLE_LIB")          # the sanitizer build (make -C oracle asan): tests/test_oracle_sanitizers.py
        if alt:
            so = pathlib.Path(alt)
        elif not so.exists():
            build()
        _LIB = C.CDLL(str(so))
        assert _LIB.oracle_struct_sizes(0) == C.sizeof(BratsParams)
        assert _LIB.oracle_struct_sizes(1) == C.sizeof(VolumeParams)
        assert _LIB.oracle_struct_sizes(2) == C.sizeof(SdfParams)
    return _LIB


def _v3(dst, src):
    a = np.asarray(src, dtype=np.float32).reshape(-1)
    for k in range(3):
        dst[k] = float(a[k])


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def brats_main(params: Dict[str, Any], vols: Sequence[Optional[np.ndarray]], labels=None, preds=None,
               ext: Optional[Dict[str, Any]] = None, *, return_aux: bool = False, rows=None):
    e = dict(DEFAULT_EXT)
    if ext:
        e.update(ext)
    P = BratsParams()
    P.width, P.height = int(params["imageSize"][0]), int(params["imageSize"][1])
    P.fovY = float(np.float32(params["fovY"]))
    for k in ("eye", "U", "V", "W", "volMin", "voxelSize", "bgColor"):
        _v3(getattr(P, k), params[k])
    for k in range(3):
        P.dims[k] = int(params["dims"][k])
    for k in ("stepSize", "nearT", "farT", "ww", "wl", "intensityAlpha", "gamma"):
        setattr(P, k, float(np.float32(params[k])))
    for k in range(4):
        P.volEnabled[k] = int(params["volEnabled"][k])
        P.volWeight[k] = float(np.float32(params["volWeight"][k]))
    P.showSeg, P.showPred = int(params["showSeg"]), int(params["showPred"])
    lut = np.asarray(params["lutColorAlpha"], dtype=np.float32).reshape(8, 4)
    for i in range(8):
        for j in range(4):
            P.lut[i][j] = float(lut[i, j])
    P.cameraMode, P.orthoHalfHeight = int(e["cameraMode"]), float(np.float32(e["orthoHalfHeight"]))
    P.shadeMode, P.specPow2 = int(e["shadeMode"]), int(e["specPow2"])
    for k in ("ka", "kd", "ks", "gradEps", "ertThreshold"):
        setattr(P, k, float(np.float32(e[k])))

    keep = []
    vp = (C.POINTER(C.c_float) * 4)()
    for m in range(4):
        v = vols[m] if m < len(vols) else None
        if v is None:
            v = np.zeros(1, np.float32)
        v = np.ascontiguousarray(v, dtype=np.float32)
        keep.append(v)
        vp[m] = _fptr(v)
    nvox = int(P.dims[0]) * int(P.dims[1]) * int(P.dims[2])
    for name, arr, shown in (("labels", labels, P.showSeg), ("preds", preds, P.showPred)):
        if shown and (arr is None or np.size(arr) < nvox):      # the C side indexes the grid unconditionally when the overlay is shown
            raise ValueError(f"show flag set but {name} is missing or smaller than the volume ({nvox} voxels)")
    for m in range(4):
        if P.volEnabled[m] and (m >= len(vols) or vols[m] is None or np.size(vols[m]) < nvox):
            raise ValueError(f"modality {m} is enabled but its grid is missing or smaller than the volume")
    lab = np.ascontiguousarray(labels if labels is not None else np.zeros(1), dtype=np.uint32)
    prd = np.ascontiguousarray(preds if preds is not None else np.zeros(1), dtype=np.uint32)
    out = np.zeros((P.height, P.width, 4), dtype=np.float32)
    r0, r1 = (0, P.height) if rows is None else rows
    stats = (C.c_uint64 * 2)()
    rc = lib().oracle_brats_main(C.byref(P), vp, lab.ctypes.data_as(C.POINTER(C.c_uint32)),
                                 prd.ctypes.data_as(C.POINTER(C.c_uint32)), _fptr(out),
                                 C.c_uint32(r0), C.c_uint32(r1), stats)
    assert rc == 0
    img = out[r0:r1]
    if return_aux:
        return img, dict(live_samples=int(stats[0]), shaded_samples=int(stats[1]))
    return img


_K2_MODES = {"u32x4": 0, "u8": 1, "f32": 2}


def volume_cs(params: Dict[str, Any], vol: np.ndarray, *, mode: str = "u32x4",
              ext: Optional[Dict[str, Any]] = None, return_aux: bool = False, rows=None):
    e = dict(DEFAULT_EXT)
    if ext:
        e.update(ext)
    P = VolumeParams()
    P.width, P.height = int(params["imageSize"][0]), int(params["imageSize"][1])
    for k in ("fovY", "stepCount", "nearPlane", "farPlane"):
        setattr(P, k, float(np.float32(params[k])))
    for k in ("eye", "U", "V", "W"):
        _v3(getattr(P, k), params[k])
    for k in range(3):
        P.volDim[k] = int(params["volDim"][k])
    P.mode = _K2_MODES[mode]
    P.cameraMode, P.orthoHalfHeight = int(e["cameraMode"]), float(np.float32(e["orthoHalfHeight"]))
    dt = {"u32x4": np.uint32, "u8": np.uint8, "f32": np.float32}[mode]
    v = np.ascontiguousarray(vol, dtype=dt)
    out = np.zeros((P.height, P.width, 4), dtype=np.float32)
    r0, r1 = (0, P.height) if rows is None else rows
    stats = (C.c_uint64 * 1)()
    rc = lib().oracle_volume_cs(C.byref(P), v.ctypes.data_as(C.c_void_p), _fptr(out),
                                C.c_uint32(r0), C.c_uint32(r1), stats)
    assert rc == 0
    img = out[r0:r1]
    if return_aux:
        return img, dict(live_samples=int(stats[0]))
    return img


def raymarch_cs(params: Dict[str, Any], eye, U, V, Wv, width: int, height: int):
    P = SdfParams()
    P.width, P.height = int(width), int(height)
    P.fovY = float(np.float32(params["fovY"]))
    P.maxSteps = int(params["maxSteps"])
    for k in ("maxDistance", "hitThreshold", "normalEps"):
        setattr(P, k, float(np.float32(params[k])))
    for k, v in (("eye", eye), ("U", U), ("V", V), ("W", Wv)):
        _v3(getattr(P, k), v)
    out = np.zeros((P.height, P.width, 4), dtype=np.float32)
    assert lib().oracle_raymarch_cs(C.byref(P), _fptr(out)) == 0
    return out
