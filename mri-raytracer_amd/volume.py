"""Load-time volume preparation: the NumPy host code either side of the kernels.

Mirrors (array-in/array-out, no NIfTI dependency):
  * ``load_nifti_float`` / ``load_seg_uint`` / world scaling / ``frame_volume`` —
    inr/viewer/brats_viewer.py:46-74,204-210,320-324
  * u8 packing, NIfTI-mask mapping and the BC4 block decode —
    scripts/volumeRendering/app.py:145-250
These run once per case on the host (as in the reference); per-frame work is all on the GPU.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def normalize_intensity(data: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(X,Y,Z) raw intensities -> (linear fp32 x-fastest buffer, normalised (X,Y,Z) array, dims).
    Percentile-1/99.5 window mapped to [0,1] (brats_viewer.py:50-65)."""
    vol = np.asarray(data, dtype=np.float32)
    lo, hi = float(np.percentile(vol, 1.0)), float(np.percentile(vol, 99.5))
    if hi <= lo:
        lo, hi = float(vol.min()), float(vol.max())
    span = max(1e-6, hi - lo)
    norm = np.clip((vol - lo) / span, 0.0, 1.0).astype(np.float32)
    return flatten_xyz(norm), norm, np.asarray(norm.shape, dtype=np.uint32)


def flatten_xyz(vol_xyz: np.ndarray) -> np.ndarray:
    """(X,Y,Z) -> 1-D with x fastest, i.e. index x + y*X + z*X*Y (brats_viewer.py:64,73)."""
    return np.ascontiguousarray(np.transpose(vol_xyz, (2, 1, 0)).reshape(-1))


def labels_to_uint(data: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Segmentation (X,Y,Z) float -> (linear uint32 buffer, dims) (brats_viewer.py:68-74)."""
    lab = np.rint(np.asarray(data, dtype=np.float32)).astype(np.uint32)
    return flatten_xyz(lab), np.asarray(lab.shape, dtype=np.uint32)


def world_frame(dims, zooms):
    """Voxel size, box origin, camera target and radius for a volume (brats_viewer.py:206-210,322-324):
    the longest axis spans 1.8 world units, centred on the origin."""
    d = np.asarray(dims).astype(np.uint32)
    k = np.float32(1.8 / float(max(d)))
    voxel_size = (np.asarray(zooms, dtype=np.float32) * k).astype(np.float32)
    extent = voxel_size * d.astype(np.float32)
    vol_min = (-0.5 * extent).astype(np.float32)
    target = (vol_min + 0.5 * extent).astype(np.float32)
    radius = float(np.linalg.norm(extent) * 0.8)
    return voxel_size, vol_min, target, radius


def pack_u8_as_u32x4(voxels_u8: np.ndarray) -> np.ndarray:
    """The reference's ``gVolumeU8`` upload: pad to a multiple of 4, widen every byte to uint32,
    view as rows of 4 (app.py:149-153).  Kept for drop-in parity; ``mode='u8'`` avoids the 4x."""
    flat = np.asarray(voxels_u8, dtype=np.uint8).reshape(-1)
    if flat.size % 4:
        flat = np.concatenate([flat, np.zeros(4 - flat.size % 4, dtype=np.uint8)])
    return flat.astype(np.uint32).reshape(-1, 4)


def mask_to_u8(data: np.ndarray, mode: str = "occupancy") -> np.ndarray:
    """NIfTI mask (X,Y,Z) -> flattened (Z,Y,X) u8 (app.py:180-197)."""
    vol = np.asarray(data, dtype=np.float32)
    if mode == "occupancy":
        u8 = np.where(vol > 0.5, 255, 0).astype(np.uint8)
    elif mode == "labels":
        u8 = np.zeros(vol.shape, dtype=np.uint8)
        for value, code in ((1.0, 85), (2.0, 170), (4.0, 255)):
            u8[np.isclose(vol, value)] = code
    else:
        raise ValueError(f"mask_mode must be 'occupancy' or 'labels', got {mode!r}")
    return flatten_xyz(u8)


def bc4_decode(bc: bytes, width: int, height: int, depth: int) -> np.ndarray:
    """BC4 (RGTC1 unorm) slices -> flattened u8 voxels (app.py:200-248).  8-byte blocks:
    two endpoints + 16 three-bit codes; r0 > r1 -> 6 interpolants, else 4 plus 0 and 255."""
    bw, bh = (width + 3) // 4, (height + 3) // 4
    want = depth * bw * bh * 8
    if len(bc) != want:
        raise RuntimeError(f"a {depth}x{height}x{width} BC4 volume is {want} bytes of 8-byte blocks, got {len(bc)}")
    blk = np.frombuffer(bc, dtype=np.uint8).reshape(depth, bh, bw, 8)
    r0 = blk[..., 0].astype(np.int32)
    r1 = blk[..., 1].astype(np.int32)
    bits = np.zeros(blk.shape[:-1], dtype=np.uint64)
    for k in range(6):
        bits |= blk[..., 2 + k].astype(np.uint64) << np.uint64(8 * k)
    six = r0 > r1
    pal = np.empty(blk.shape[:-1] + (8,), dtype=np.int32)
    pal[..., 0], pal[..., 1] = r0, r1
    for i in range(1, 7):
        a = ((7 - i) * r0 + i * r1 + 3) // 7
        if i <= 4:
            b = ((5 - i) * r0 + i * r1 + 2) // 5
        else:
            b = np.full_like(r0, 0 if i == 5 else 255)
        pal[..., i + 1] = np.where(six, a, b)
    codes = (bits[..., None] >> (np.arange(16, dtype=np.uint64) * np.uint64(3))) & np.uint64(7)
    texels = np.take_along_axis(pal, codes.astype(np.int64), axis=-1).astype(np.uint8)   # (D,bh,bw,16)
    img = texels.reshape(depth, bh, bw, 4, 4).transpose(0, 1, 3, 2, 4).reshape(depth, bh * 4, bw * 4)
    return np.ascontiguousarray(img[:, :height, :width]).reshape(-1)


def bc4_decode_device(bc, width: int, height: int, depth: int, stream=None):
    """The same decode on the GPU (csrc/grid_ops.hip): ``bc`` is the block stream as bytes, a NumPy uint8
    array or a device uint8 tensor; returns the flattened (D*H*W,) device uint8 tensor, ready to be bound as
    ``gVolumeU8`` with ``mode='u8'``."""
    import ctypes as C
    import torch
    from . import _lib
    if not torch.cuda.is_available():
        raise RuntimeError("bc4_decode_device needs an MI355X; volume.bc4_decode is the host decode")
    bw, bh = (width + 3) // 4, (height + 3) // 4
    want = depth * bw * bh * 8
    if isinstance(bc, (bytes, bytearray, memoryview)):
        bc = np.frombuffer(bytes(bc), dtype=np.uint8).copy()
    t = torch.as_tensor(bc)
    if t.dtype != torch.uint8 or t.numel() != want:
        raise RuntimeError(f"a {depth}x{height}x{width} BC4 volume is {want} bytes of 8-byte blocks, got {t.numel()}")
    t = t.reshape(-1).cuda().contiguous()
    out = torch.empty(depth * height * width, dtype=torch.uint8, device=t.device)
    s = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
    _lib.check(_lib.lib().mrirt_bc4_decode(C.c_void_p(t.data_ptr()), width, height, depth, C.c_void_p(out.data_ptr()), s),
               "mrirt_bc4_decode")
    return out


def zscore_nonzero(arr: np.ndarray) -> np.ndarray:
    """Per-modality z-score over non-zero voxels, sigma + 1e-6 (brats_viewer.py:281-287)."""
    a = np.asarray(arr, dtype=np.float32)
    nz = a != 0
    if nz.any():
        a = (a - a[nz].mean()) / (a[nz].std() + 1e-6)
    return a
