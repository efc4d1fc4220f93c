"""Orbit camera producing the (eye, U, V, W) frame the kernels consume.

Host-side mirror of the reference's two ``OrbitalCamera`` classes — same constructor keywords,
attributes and methods, so viewer-style code ports unchanged:
  * Y-up variant:      scripts/raymarch/camera.py:9-114   (``world_up=None`` here)
  * world-up variant:  inr/viewer/camera.py:8-129         (``world_up=<vector>``)
Trigonometry is done in Python floats (fp64) and stored as float32, as in the reference; the
results are checked bit-for-bit against goldens captured from the reference classes
(tests/golden/camera_*.npz).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np

_F32 = np.float32


class OrbitalCamera:
    _STATE = ("target", "radius", "phi", "theta", "min_radius", "max_radius", "min_phi", "max_phi")

    def __init__(self, initial_target: Optional[np.ndarray] = None, initial_radius: float = 2.0,
                 initial_phi: float = math.pi * 0.5, initial_theta: float = 0.0,
                 min_radius: float = 0.1, max_radius: float = 100.0,
                 min_phi: float = 0.01, max_phi: float = math.pi - 0.01,
                 aspect: float = 16.0 / 9.0, fovY_radians: float = math.radians(55.0),
                 near: float = 0.1, far: float = 1000.0, world_up: Optional[np.ndarray] = None):
        tgt = np.zeros(3, dtype=_F32) if initial_target is None else np.asarray(initial_target).astype(_F32)
        self._initial = dict(target=tgt, radius=float(initial_radius), phi=float(initial_phi),
                             theta=float(initial_theta), min_radius=float(min_radius),
                             max_radius=float(max_radius), min_phi=float(min_phi), max_phi=float(max_phi))
        self.fovY_radians, self.aspect = float(fovY_radians), float(aspect)
        self.near, self.far = float(near), float(far)
        # None selects the fixed +Y camera of scripts/raymarch (which has no world_up attribute)
        self._y_up = world_up is None
        self.world_up = (np.array([0.0, 1.0, 0.0], dtype=_F32) if world_up is None
                         else np.asarray(world_up).astype(_F32))
        self.reset()

    # -- state ---------------------------------------------------------------------------
    def reset(self) -> None:
        for k in self._STATE:
            v = self._initial[k]
            setattr(self, k, v.copy() if isinstance(v, np.ndarray) else v)

    def orbit(self, d_theta: float, d_phi: float) -> None:
        self.theta += float(d_theta)
        self.phi = min(self.max_phi, max(self.min_phi, self.phi + float(d_phi)))

    def zoom(self, factor: float) -> None:
        self.radius = min(self.max_radius, max(self.min_radius, self.radius * float(factor)))

    def pan(self, dx: float, dy: float, viewport_height: Optional[float] = None) -> None:
        _, right, up, _ = self.get_basis()
        if self._y_up or viewport_height is None or viewport_height <= 0:
            pixels = 720.0
        else:
            pixels = max(1.0, float(viewport_height))
        world_per_px = 2.0 * self.radius * math.tan(max(1e-3, self.fovY_radians * 0.5)) / pixels
        self.target = (self.target - right * (float(dx) * world_per_px)
                       + up * (float(dy) * world_per_px)).astype(_F32)

    def set_fov_degrees(self, fov_deg: float) -> None:
        self.fovY_radians = math.radians(float(fov_deg))

    def set_aspect(self, aspect: float) -> None:
        self.aspect = float(aspect)

    # -- frame ---------------------------------------------------------------------------
    def _base_frame(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        up = self.world_up
        ref = np.array([0.0, 0.0, 1.0], dtype=_F32)
        if abs(float(np.dot(up, ref))) > 0.999:
            ref = np.array([1.0, 0.0, 0.0], dtype=_F32)
        r = np.cross(ref, up)
        rn = float(np.linalg.norm(r))
        if rn < 1e-6:
            r, rn = np.array([1.0, 0.0, 0.0], dtype=_F32), 1.0
        r = (r / rn).astype(_F32)
        f = np.cross(up, r).astype(_F32)
        fn = float(np.linalg.norm(f))
        if fn > 0:
            f = (f / fn).astype(_F32)
        return r, f, up

    def get_eye_position(self) -> np.ndarray:
        s, c = math.sin(self.phi), math.cos(self.phi)
        if self._y_up:
            t = self.target
            return np.array([t[0] + self.radius * s * math.cos(self.theta),
                             t[1] + self.radius * c,
                             t[2] + self.radius * s * math.sin(self.theta)], dtype=_F32)
        r, f, u = self._base_frame()
        offset = (s * math.cos(self.theta)) * r + (s * math.sin(self.theta)) * f + c * u
        return (self.target + self.radius * offset.astype(_F32)).astype(_F32)

    def get_basis(self):
        """-> eye, right (U), up (V), forward (W), float32[3] each."""
        eye = self.get_eye_position()
        look = self.target - eye
        ln = float(np.linalg.norm(look))
        forward = np.array([0.0, 0.0, -1.0], dtype=_F32) if ln < 1e-6 else (look / ln).astype(_F32)
        right = np.cross(forward, self.world_up)
        rn = float(np.linalg.norm(right))
        if rn < 1e-6:
            if self._y_up:
                right = np.cross(forward, np.array([0.0, 0.0, 1.0], dtype=_F32))
            else:
                right = self._base_frame()[0]
            rn = float(np.linalg.norm(right))
        if rn > 0:
            right = (right / rn).astype(_F32)
        up = np.cross(right, forward).astype(_F32)
        if not self._y_up and float(np.dot(up, self.world_up)) < 0.0:
            up, right = -up, -right
        return eye.astype(_F32), right, up, forward
