"""Deterministic synthetic scenes for tests and bench (no dataset ships with the build).

Definitions follow SURVEY.md section 8(d): a soft ball with a trigonometric ripple plus 5 %
uniform noise, an orbit camera at (r=3, phi=80deg, theta=25deg), fovY 45deg, the viewer's
transfer-function defaults (inr/viewer/brats_viewer.py:126-144) and the BASELINE configs
C1..C4.  Everything is generated in NumPy fp32 on the host so the oracle and the GPU see the
very same bytes.
"""
from __future__ import annotations

import math
from typing import Any, Dict, Optional, Tuple

import numpy as np

from .camera import OrbitalCamera

# LUT of the viewer (brats_viewer.py:138-143)
VIEWER_LUT = np.array([[0.0, 0.0, 0.0, 0.0], [0.0, 0.4, 1.0, 0.9], [0.0, 0.8, 0.0, 0.7],
                       [1.0, 0.1, 0.1, 0.9], [1.0, 0.1, 0.1, 0.9], [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]],
                      dtype=np.float32)


def synth_volume(n: int, seed: int = 1234, phase: float = 0.0, dims: Optional[Tuple[int, int, int]] = None) -> np.ndarray:
    """fp32 linear (x fastest) volume in [0,1]; ``dims=(X,Y,Z)`` for non-cubic grids."""
    X, Y, Z = dims if dims is not None else (n, n, n)
    rng = np.random.default_rng(seed)
    xs = np.linspace(-1.0, 1.0, X, dtype=np.float32)
    ys = np.linspace(-1.0, 1.0, Y, dtype=np.float32)
    zs = np.linspace(-1.0, 1.0, Z, dtype=np.float32)
    out = np.empty((Z, Y, X), dtype=np.float32)
    sx = np.sin(np.float32(8.0) * xs + np.float32(phase))[None, :]
    cy = np.cos(np.float32(6.0) * ys)[:, None]
    for k in range(Z):            # slice by slice: bounded temporaries at 512^3
        r = np.sqrt(xs[None, :] ** 2 + ys[:, None] ** 2 + zs[k] ** 2)
        ripple = np.float32(0.75) + np.float32(0.25) * sx * cy * np.sin(np.float32(5.0) * zs[k])
        v = np.clip((np.float32(1.0) - r) * ripple, 0.0, 1.0)
        v = v + np.float32(0.05) * rng.random((Y, X), dtype=np.float32)
        out[k] = np.clip(v, 0.0, 1.0)
    return out.reshape(-1)


def synth_labels(n: int, dims: Optional[Tuple[int, int, int]] = None) -> np.ndarray:
    """uint32 linear label grid: nested spheres r<0.25 -> 3, <0.4 -> 1, <0.55 -> 2, else 0."""
    X, Y, Z = dims if dims is not None else (n, n, n)
    xs = np.linspace(-1.0, 1.0, X, dtype=np.float32)
    ys = np.linspace(-1.0, 1.0, Y, dtype=np.float32)
    zs = np.linspace(-1.0, 1.0, Z, dtype=np.float32)
    r = np.sqrt(xs[None, None, :] ** 2 + ys[None, :, None] ** 2 + zs[:, None, None] ** 2)
    lab = np.zeros((Z, Y, X), dtype=np.uint32)
    lab[r < 0.55] = 2
    lab[r < 0.4] = 1
    lab[r < 0.25] = 3
    return lab.reshape(-1)


def synth_u8_volume(n: int, seed: int = 1234) -> np.ndarray:
    return np.rint(synth_volume(n, seed) * 255.0).astype(np.uint8)


def bench_camera(radius: float = 3.0, phi_deg: float = 80.0, theta_deg: float = 25.0, world_up=None) -> OrbitalCamera:
    return OrbitalCamera(initial_radius=radius, initial_phi=math.radians(phi_deg),
                         initial_theta=math.radians(theta_deg), world_up=world_up)


def brats_scene(n: int, image: int, steps: int, *, channels: int = 1, intensity_alpha: float = 0.4,
                show_seg: bool = False, show_pred: bool = False, image_hw: Optional[Tuple[int, int]] = None,
                dims: Optional[Tuple[int, int, int]] = None, camera: Optional[OrbitalCamera] = None,
                fov_deg: float = 45.0) -> Dict[str, Any]:
    """gParams dict for K1 on the synthetic scene: box of 1.8 world units centred at the origin,
    stepSize = 1.8*sqrt(3)/steps so no ray takes more than ``steps`` samples."""
    X, Y, Z = dims if dims is not None else (n, n, n)
    cam = camera or bench_camera()
    eye, U, V, W = cam.get_basis()
    H, Wd = image_hw if image_hw is not None else (image, image)
    vox = np.float32(1.8 / max(X, Y, Z))
    voxel = np.array([vox, vox, vox], dtype=np.float32)
    ext = voxel * np.array([X, Y, Z], dtype=np.float32)
    return {
        "imageSize": (Wd, H), "fovY": math.radians(fov_deg),
        "eye": eye, "U": U, "V": V, "W": W,
        "volMin": (-0.5 * ext).astype(np.float32), "voxelSize": voxel,
        "dims": np.array([X, Y, Z], dtype=np.uint32),
        "stepSize": float(np.float32(1.8 * math.sqrt(3.0) / steps)), "nearT": 0.0, "farT": 0.0,
        "bgColor": np.zeros(3, dtype=np.float32),
        "volEnabled": tuple(np.uint32(1 if m < channels else 0) for m in range(4)),
        "volWeight": (1.0, 1.0, 1.0, 1.0),
        "ww": 1.0, "wl": 0.5, "intensityAlpha": float(intensity_alpha),
        "gamma": 1.0, "gradBoost": 1.5, "gradScale": 1.0,
        "showSeg": 1 if show_seg else 0, "showPred": 1 if show_pred else 0,
        "lutColorAlpha": [tuple(map(float, row)) for row in VIEWER_LUT.tolist()],
    }


def volume_scene(n: int, image: int, steps: int, near: float = 1.5, far: float = 4.5,
                 fov_deg: float = 45.0, camera: Optional[OrbitalCamera] = None,
                 dims: Optional[Tuple[int, int, int]] = None) -> Dict[str, Any]:
    """gParams dict for K2: cube (-1,1)^3 between the near and far planes."""
    X, Y, Z = dims if dims is not None else (n, n, n)
    cam = camera or bench_camera()
    eye, U, V, W = cam.get_basis()
    return {"imageSize": (np.uint32(image), np.uint32(image)), "fovY": np.float32(math.radians(fov_deg)),
            "stepCount": np.float32(steps), "nearPlane": np.float32(near), "farPlane": np.float32(far),
            "eye": eye, "U": U, "V": V, "W": W,
            "volDim": (np.uint32(X), np.uint32(Y), np.uint32(Z))}


def sdf_scene(fov_deg: float = 45.0, max_steps: int = 128, camera: Optional[OrbitalCamera] = None):
    """K3 defaults of scripts/raymarch/app.py:52-56."""
    cam = camera or bench_camera(radius=2.0)
    eye, U, V, W = cam.get_basis()
    params = {"fovY": np.float32(math.radians(fov_deg)), "maxSteps": np.uint32(max_steps),
              "maxDistance": np.float32(10.0), "hitThreshold": np.float32(1e-3), "normalEps": np.float32(1e-3)}
    return params, eye, U, V, W


# Blinn-Phong extension constants of BASELINE config 3
SHADE_EXT = dict(shadeMode=1, ka=0.3, kd=0.6, ks=0.3, specPow2=5, gradEps=1e-6)

# BASELINE.json configs (index -> description of the synthetic stand-in)
CONFIGS = {
    "C1": dict(kernel="volume_f32_ortho", n=128, image=256, steps=64),
    "C2": dict(kernel="brats", n=256, image=512, steps=256, channels=4, show_seg=True, intensity_alpha=0.4),
    "C3": dict(kernel="brats", n=512, image=1024, steps=512, channels=1, shade=True, intensity_alpha=16.0),
    "C4": dict(kernel="brats", n=512, image=2048, steps=512, channels=1, shade=True, intensity_alpha=16.0, gpus=8),
}
