"""Minimal NIfTI-1 (.nii / .nii.gz) reader and writer — the ingest side of the viewer.

The reference reads volumes with nibabel (``nib.load(path).get_fdata(dtype=np.float32)`` and
``header.get_zooms()``, inr/viewer/brats_viewer.py:46-74), which is not a dependency here.  This
module reads the single-file NIfTI-1 layout BraTS ships (348-byte header, data at ``vox_offset``,
x fastest on disk) and returns exactly what those two calls return: an (X, Y, Z) float32 array
with ``scl_slope``/``scl_inter`` applied, and the first three ``pixdim`` as zooms.
``load_nifti_float`` / ``load_seg_uint`` then mirror the viewer's functions of the same names.
"""
from __future__ import annotations

import gzip
import pathlib
import struct
from typing import Tuple

import numpy as np

from . import volume

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64,
           256: np.int8, 512: np.uint16, 768: np.uint32}
_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}


def _read_all(path: pathlib.Path) -> bytes:
    raw = path.read_bytes()
    return gzip.decompress(raw) if raw[:2] == b"\x1f\x8b" else raw


def read_nifti(path) -> Tuple[np.ndarray, np.ndarray]:
    """-> (data (X,Y,Z[,T...]) float32 with scaling applied, zooms float32[3])."""
    path = pathlib.Path(path)
    buf = _read_all(path)
    if len(buf) < 352:
        raise ValueError(f"{path}: too short for a NIfTI-1 header")
    end = "<" if struct.unpack_from("<i", buf, 0)[0] == 348 else ">"
    if struct.unpack_from(end + "i", buf, 0)[0] != 348:
        raise ValueError(f"{path}: sizeof_hdr is not 348 (not NIfTI-1)")
    magic = buf[344:348]
    if magic not in (b"n+1\0", b"ni1\0"):
        raise ValueError(f"{path}: bad NIfTI magic {magic!r}")
    if magic == b"ni1\0":
        raise ValueError(f"{path}: two-file NIfTI (.hdr/.img) is not supported")
    dim = struct.unpack_from(end + "8h", buf, 40)
    ndim = dim[0]
    if not 1 <= ndim <= 7:
        raise ValueError(f"{path}: dim[0] = {ndim}")
    shape = tuple(int(d) for d in dim[1:1 + ndim])
    datatype, _bitpix = struct.unpack_from(end + "2h", buf, 70)
    if datatype not in _DTYPES:
        raise ValueError(f"{path}: unsupported datatype code {datatype}")
    pixdim = struct.unpack_from(end + "8f", buf, 76)
    vox_offset, slope, inter = struct.unpack_from(end + "3f", buf, 108)
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(end)
    count = int(np.prod(shape))
    off = int(vox_offset) if vox_offset >= 352 else 352
    if len(buf) < off + count * dt.itemsize:
        raise ValueError(f"{path}: data truncated ({len(buf) - off} bytes for {count} voxels)")
    data = np.frombuffer(buf, dtype=dt, count=count, offset=off).reshape(shape[::-1])
    data = np.transpose(data, tuple(range(len(shape) - 1, -1, -1)))          # x fastest on disk -> (X, Y, Z, ...)
    out = data.astype(np.float32)
    if slope != 0 and np.isfinite(slope) and not (slope == 1.0 and inter == 0.0):
        out = out * np.float32(slope) + np.float32(inter)
    zooms = np.array([abs(p) for p in pixdim[1:4]], dtype=np.float32)
    return out, zooms


def write_nifti(path, data: np.ndarray, zooms=(1.0, 1.0, 1.0), slope: float = 1.0, inter: float = 0.0) -> None:
    """Write a little-endian single-file NIfTI-1 (used by tests and tools to make fixtures)."""
    path = pathlib.Path(path)
    arr = np.asarray(data)
    if arr.dtype not in _CODES:
        arr = arr.astype(np.float32)
    hdr = bytearray(352)
    struct.pack_into("<i", hdr, 0, 348)
    dims = [arr.ndim] + list(arr.shape) + [1] * (7 - arr.ndim)
    struct.pack_into("<8h", hdr, 40, *dims)
    struct.pack_into("<2h", hdr, 70, _CODES[arr.dtype], arr.dtype.itemsize * 8)
    struct.pack_into("<8f", hdr, 76, 1.0, *[float(z) for z in zooms], 1.0, 1.0, 1.0, 1.0)
    struct.pack_into("<3f", hdr, 108, 352.0, float(slope), float(inter))
    hdr[344:348] = b"n+1\0"
    payload = bytes(hdr) + np.ascontiguousarray(np.transpose(arr, tuple(range(arr.ndim - 1, -1, -1)))).astype(
        arr.dtype.newbyteorder("<")).tobytes()
    path.write_bytes(gzip.compress(payload, 1) if path.suffix == ".gz" else payload)


def load_nifti_float(path):
    """inr/viewer/brats_viewer.py:46-65: -> (linear fp32 x-fastest, normalised (X,Y,Z), dims u32[3], zooms)."""
    data, zooms = read_nifti(path)
    linear, norm, dims = volume.normalize_intensity(data[..., 0] if data.ndim == 4 else data)
    return linear, norm, dims, zooms


def load_seg_uint(path):
    """inr/viewer/brats_viewer.py:68-74: -> (linear uint32 x-fastest, dims u32[3], zooms)."""
    data, zooms = read_nifti(path)
    linear, dims = volume.labels_to_uint(data[..., 0] if data.ndim == 4 else data)
    return linear, dims, zooms
