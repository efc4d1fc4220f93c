"""CPU oracle (NumPy, fp32) for the MRI ray-march hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, operation by operation, the three Slang compute shaders of
klukaszek/MRI-RayTracer and the NumPy/JAX host code either side of them.  It is the
*checker*: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  Nothing under ``mri-raytracer_amd/`` does.

Reference files followed (paths relative to the reference checkout):
  K1  inr/viewer/brats_rt.slang:12-168            -> brats_main()
  K2  scripts/volumeRendering/volume_render.slang:9-65,104-148 -> volume_cs()
  K3  scripts/raymarch/raymarch.slang:7-99        -> raymarch_cs()
  cam scripts/raymarch/camera.py:61-88, inr/viewer/camera.py:62-123 -> OrbitalCamera*
  prep inr/viewer/brats_viewer.py:46-74,204-210,320-324;
       scripts/volumeRendering/app.py:145-250     -> prep functions
  INR inr/inr/model.py:11-50,119-141,217-301;  notebooks/neumors_inr.ipynb:1150-1178

Parity status: the Slang kernels cannot be compiled here (slangpy absent) and the
reference has no tests or golden images, so image-level parity is *unpinned by the
reference*; the pins are (i) camera and INR goldens produced by importing the reference's
own Python (tests/golden/make_goldens.py) and (ii) analytic known answers
(tests/test_oracle_known_answers.py).

Arithmetic conventions (where the Slang text leaves them to the backend):
  * every operation is IEEE fp32, unfused (no FMA), evaluated left to right as written;
  * lerp(a,b,t) = a + t*(b-a); saturate(x) = fmin(fmax(x,0),1) (NaN -> 0, as HLSL);
  * transcendentals (tan, exp, pow, atan2) are *correctly rounded* fp32, obtained by
    evaluating in fp64 and rounding once; sqrt and divide are IEEE;
  * normalize(v) = v / sqrt((x*x + y*y) + z*z);
  * round() in sampleLabel is round-half-away-from-zero (Metal ``round``; the reference's
    authors ran Slang->Metal: docs/showcase-plan.md:5).
"""
from __future__ import annotations

import json
import math
import pathlib
from typing import Any, Dict, Optional, Sequence, Tuple

import numpy as np

F = np.float32
_ONE = F(1.0)
_ZERO = F(0.0)


# ----------------------------------------------------------------------------------------
# fp32 helpers
# ----------------------------------------------------------------------------------------
def _f(x) -> np.ndarray:
    return np.asarray(x, dtype=np.float32)


def _exp(x):
    """Correctly rounded fp32 exp (fp64 evaluation, one rounding)."""
    return np.exp(np.asarray(x, dtype=np.float64)).astype(np.float32)


def _pow(x, y):
    return np.power(np.asarray(x, dtype=np.float64), np.float64(y)).astype(np.float32)


def _tan(x):
    return F(math.tan(float(F(x))))


def _sat(x):
    return np.fmin(np.fmax(x, _ZERO), _ONE)


def _lerp(a, b, t):
    return a + t * (b - a)


def _dot3(ax, ay, az, bx, by, bz):
    return (ax * bx + ay * by) + az * bz


def _normalize3(x, y, z):
    n = np.sqrt(_dot3(x, y, z, x, y, z))
    return x / n, y / n, z / n


def _round_half_away(x):
    """roundf() for x >= 0 (x - floor(x) is exact in fp32)."""
    fl = np.floor(x)
    return np.where((x - fl) >= F(0.5), fl + _ONE, fl)


def _vec3(v) -> Tuple[np.float32, np.float32, np.float32]:
    a = np.asarray(v, dtype=np.float32).reshape(-1)
    return F(a[0]), F(a[1]), F(a[2])


# ----------------------------------------------------------------------------------------
# Ray generation
# ----------------------------------------------------------------------------------------
def make_primary(width: int, height: int, fovY, eye, U, V, Wv, *, k3_aspect: bool = False):
    """Perspective primary rays for every pixel.  brats_rt.slang:36-46
    (raymarch.slang:45-58 when ``k3_aspect``: aspect = dim.x/dim.y without the max()).

    Returns (ox,oy,oz) scalars and (dx,dy,dz) arrays of shape (H,W); row 0 is the top row.
    """
    px = np.arange(width, dtype=np.float32)[None, :]
    py = np.arange(height, dtype=np.float32)[:, None]
    dimx, dimy = F(width), F(height)
    ndcx = (px + F(0.5)) / dimx
    ndcy = (py + F(0.5)) / dimy
    uvx = ndcx * F(2.0) - _ONE
    uvy = ndcy * F(2.0) - _ONE
    f = _ONE / _tan(F(0.5) * F(fovY))
    aspect = dimx / dimy if k3_aspect else dimx / max(_ONE, dimy)
    cx = np.broadcast_to(uvx * aspect / f, (height, width)).astype(np.float32)
    cy = np.broadcast_to(-uvy / f, (height, width)).astype(np.float32)
    cz = np.ones((height, width), dtype=np.float32)
    cx, cy, cz = _normalize3(cx, cy, cz)
    Ux, Uy, Uz = _vec3(U)
    Vx, Vy, Vz = _vec3(V)
    Wx, Wy, Wz = _vec3(Wv)
    dx = (cx * Ux + cy * Vx) + cz * Wx
    dy = (cx * Uy + cy * Vy) + cz * Wy
    dz = (cx * Uz + cy * Vz) + cz * Wz
    dx, dy, dz = _normalize3(dx, dy, dz)
    return _vec3(eye), (dx, dy, dz)


def make_ortho(width: int, height: int, half_height, eye, U, V, Wv):
    """BUILD-DEFINED extension (no reference counterpart):
    o = eye + U*(uv.x*h*aspect) + V*(-uv.y*h), d = W."""
    px = np.arange(width, dtype=np.float32)[None, :]
    py = np.arange(height, dtype=np.float32)[:, None]
    dimx, dimy = F(width), F(height)
    uvx = ((px + F(0.5)) / dimx) * F(2.0) - _ONE
    uvy = ((py + F(0.5)) / dimy) * F(2.0) - _ONE
    h = F(half_height)
    aspect = dimx / max(_ONE, dimy)
    sx = np.broadcast_to(uvx * h * aspect, (height, width)).astype(np.float32)
    sy = np.broadcast_to(-uvy * h, (height, width)).astype(np.float32)
    ex, ey, ez = _vec3(eye)
    Ux, Uy, Uz = _vec3(U)
    Vx, Vy, Vz = _vec3(V)
    Wx, Wy, Wz = _vec3(Wv)
    ox = (ex + Ux * sx) + Vx * sy
    oy = (ey + Uy * sx) + Vy * sy
    oz = (ez + Uz * sx) + Vz * sy
    ones = np.ones((height, width), dtype=np.float32)
    return (ox, oy, oz), (ones * Wx, ones * Wy, ones * Wz)


# ----------------------------------------------------------------------------------------
# K1: brats_main
# ----------------------------------------------------------------------------------------
def _sample_linear(buf, qx, qy, qz, X, Y, Z):
    """sampleLinear, brats_rt.slang:60-76.  Returns value and (i,f) for reuse."""
    cx = np.minimum(np.maximum(qx, _ZERO), F(X) - F(1.001))
    cy = np.minimum(np.maximum(qy, _ZERO), F(Y) - F(1.001))
    cz = np.minimum(np.maximum(qz, _ZERO), F(Z) - F(1.001))
    fx0, fy0, fz0 = np.floor(cx), np.floor(cy), np.floor(cz)
    ix, iy, iz = fx0.astype(np.int64), fy0.astype(np.int64), fz0.astype(np.int64)
    fx, fy, fz = cx - fx0, cy - fy0, cz - fz0
    sY, sZ = X, X * Y
    b = ix + iy * sY + iz * sZ
    c000, c100 = buf[b], buf[b + 1]
    c010, c110 = buf[b + sY], buf[b + sY + 1]
    c001, c101 = buf[b + sZ], buf[b + sZ + 1]
    c011, c111 = buf[b + sZ + sY], buf[b + sZ + sY + 1]
    v = _lerp(_lerp(_lerp(c000, c100, fx), _lerp(c010, c110, fx), fy),
              _lerp(_lerp(c001, c101, fx), _lerp(c011, c111, fx), fy), fz)
    return v, (ix, iy, iz, fx, fy, fz)


def _lattice_gradient(buf, ix, iy, iz, fx, fy, fz, X, Y, Z):
    """BUILD-DEFINED extension: trilinear interpolation of lattice central differences
    (== central difference of the trilinear field with h = 1 voxel in the interior).
    For each axis the 8 corner differences buf[clamp(c+e)] - buf[clamp(c-e)] are blended
    with the same nested lerp and the same fractions as sampleLinear."""
    out = []
    for axis in range(3):
        d = {}
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    cx, cy, cz = ix + dx, iy + dy, iz + dz
                    if axis == 0:
                        p = np.minimum(cx + 1, X - 1) + cy * X + cz * X * Y
                        m = np.maximum(cx - 1, 0) + cy * X + cz * X * Y
                    elif axis == 1:
                        p = cx + np.minimum(cy + 1, Y - 1) * X + cz * X * Y
                        m = cx + np.maximum(cy - 1, 0) * X + cz * X * Y
                    else:
                        p = cx + cy * X + np.minimum(cz + 1, Z - 1) * X * Y
                        m = cx + cy * X + np.maximum(cz - 1, 0) * X * Y
                    d[(dx, dy, dz)] = buf[p] - buf[m]
        g = _lerp(_lerp(_lerp(d[0, 0, 0], d[1, 0, 0], fx), _lerp(d[0, 1, 0], d[1, 1, 0], fx), fy),
                  _lerp(_lerp(d[0, 0, 1], d[1, 0, 1], fx), _lerp(d[0, 1, 1], d[1, 1, 1], fx), fy), fz)
        out.append(g)
    return out


def _sample_label(buf, qx, qy, qz, X, Y, Z):
    """sampleLabel, brats_rt.slang:78-83."""
    ix = _round_half_away(np.minimum(np.maximum(qx, _ZERO), F(X) - _ONE)).astype(np.int64)
    iy = _round_half_away(np.minimum(np.maximum(qy, _ZERO), F(Y) - _ONE)).astype(np.int64)
    iz = _round_half_away(np.minimum(np.maximum(qz, _ZERO), F(Z) - _ONE)).astype(np.int64)
    return buf[ix + iy * X + iz * X * Y]


DEFAULT_EXT = dict(
    cameraMode=0,          # 0 perspective (reference), 1 orthographic (extension)
    orthoHalfHeight=1.1,
    shadeMode=0,           # 0 off (reference), 1 lattice-gradient Blinn-Phong headlight
    ka=0.3, kd=0.6, ks=0.3,
    specPow2=5,            # specular exponent = 2**specPow2 (5 -> 32), by repeated squaring
    gradEps=1e-6,
    ertThreshold=0.01,     # the reference's hard-coded 0.01 (brats_rt.slang:117)
)


def brats_main(params: Dict[str, Any], vols: Sequence[Optional[np.ndarray]],
               labels: Optional[np.ndarray] = None, preds: Optional[np.ndarray] = None,
               ext: Optional[Dict[str, Any]] = None, *, return_aux: bool = False,
               rows: Optional[Tuple[int, int]] = None):
    """K1: inr/viewer/brats_rt.slang:85-168 for every pixel.

    ``params`` is the reference's ``gParams`` dict (brats_viewer.py:405-426), ``vols`` the
    four ``gIntensity*`` linear fp32 buffers (x fastest; None/dummy when disabled),
    ``labels``/``preds`` the linear uint32 buffers.  Returns float32 (H,W,4).
    ``rows=(r0,r1)`` renders only that band of rows (same values as the full frame).
    """
    e = dict(DEFAULT_EXT)
    if ext:
        e.update(ext)
    Wd, Hd = int(params["imageSize"][0]), int(params["imageSize"][1])
    X, Y, Z = (int(v) for v in params["dims"])
    bminx, bminy, bminz = _vec3(params["volMin"])
    vsx, vsy, vsz = _vec3(params["voxelSize"])
    bmaxx = bminx + vsx * F(X)
    bmaxy = bminy + vsy * F(Y)
    bmaxz = bminz + vsz * F(Z)
    step = F(params["stepSize"])
    nearT, farT = F(params["nearT"]), F(params["farT"])
    bg = _f(params["bgColor"]).reshape(3)
    en = [int(v) for v in params["volEnabled"]]
    wt = [F(v) for v in params["volWeight"]]
    ww, wl, ia = F(params["ww"]), F(params["wl"]), F(params["intensityAlpha"])
    gamma = F(params["gamma"])
    showSeg, showPred = int(params["showSeg"]), int(params["showPred"])
    lut = _f(params["lutColorAlpha"]).reshape(8, 4)
    ert = F(e["ertThreshold"])

    if int(e["cameraMode"]) == 0:
        (ox, oy, oz), (dx, dy, dz) = make_primary(Wd, Hd, params["fovY"], params["eye"],
                                                  params["U"], params["V"], params["W"])
        ox = np.full((Hd, Wd), ox, np.float32)
        oy = np.full((Hd, Wd), oy, np.float32)
        oz = np.full((Hd, Wd), oz, np.float32)
    else:
        (ox, oy, oz), (dx, dy, dz) = make_ortho(Wd, Hd, e["orthoHalfHeight"], params["eye"],
                                                params["U"], params["V"], params["W"])
    r0, r1 = (0, Hd) if rows is None else rows
    sl = slice(r0, r1)
    ox, oy, oz, dx, dy, dz = (a[sl].reshape(-1) for a in (ox, oy, oz, dx, dy, dz))
    n = ox.size

    eps = F(1e-6)
    ddx = np.where(np.abs(dx) < eps, eps, dx)
    ddy = np.where(np.abs(dy) < eps, eps, dy)
    ddz = np.where(np.abs(dz) < eps, eps, dz)
    rx, ry, rz = _ONE / ddx, _ONE / ddy, _ONE / ddz
    t0x, t1x = (bminx - ox) * rx, (bmaxx - ox) * rx
    t0y, t1y = (bminy - oy) * ry, (bmaxy - oy) * ry
    t0z, t1z = (bminz - oz) * rz, (bmaxz - oz) * rz
    tmin = np.maximum(np.maximum(np.minimum(t0x, t1x), np.minimum(t0y, t1y)), np.minimum(t0z, t1z))
    tmax = np.minimum(np.minimum(np.maximum(t0x, t1x), np.maximum(t0y, t1y)), np.maximum(t0z, t1z))
    hit = tmax >= np.maximum(tmin, _ZERO)
    t0 = np.maximum(tmin, max(_ZERO, nearT))
    t1 = np.minimum(tmax, farT) if farT > 0 else tmax
    live = hit & ~(t1 <= t0)

    C = np.empty((n, 3), dtype=np.float32)
    C[:] = bg
    T = np.ones(n, dtype=np.float32)
    t = t0.astype(np.float32).copy()
    nsteps = np.zeros(n, dtype=np.int32)
    nshaded = np.zeros(n, dtype=np.int32)
    fragile = np.zeros(n, dtype=bool)

    idx = np.nonzero(live)[0]
    # loop condition is evaluated before every iteration (while t<t1 && T>0.01)
    while idx.size:
        tt, TT = t[idx], T[idx]
        if return_aux:
            # knife-edge detector: a 1e-5 relative perturbation of T, or 4 ulp of t, flips the test
            near_T = np.abs(TT - ert) <= ert * F(1e-5)
            near_t = np.abs(tt - t1[idx]) <= F(4.0) * np.spacing(np.abs(t1[idx]))
            fragile[idx] |= near_T | near_t
        go = (tt < t1[idx]) & (TT > ert)
        idx = idx[go]
        if not idx.size:
            break
        tt = t[idx]
        ddx_, ddy_, ddz_ = dx[idx], dy[idx], dz[idx]
        px_ = ox[idx] + tt * ddx_
        py_ = oy[idx] + tt * ddy_
        pz_ = oz[idx] + tt * ddz_
        qx = (px_ - bminx) / vsx
        qy = (py_ - bminy) / vsy
        qz = (pz_ - bminz) / vsz

        v = np.zeros(idx.size, dtype=np.float32)
        wsum = _ZERO
        shade_on = int(e["shadeMode"]) != 0
        g = [np.zeros(idx.size, dtype=np.float32) for _ in range(3)] if shade_on else None
        for m in range(4):
            if en[m] != 0:
                s, (ix, iy, iz, fx, fy, fz) = _sample_linear(vols[m], qx, qy, qz, X, Y, Z)
                v = v + s * wt[m]
                wsum = wsum + wt[m]
                if shade_on:
                    gm = _lattice_gradient(vols[m], ix, iy, iz, fx, fy, fz, X, Y, Z)
                    for a in range(3):
                        g[a] = g[a] + gm[a] * wt[m]
        if wsum > 0:
            v = v / wsum          # the gradient is used for its direction only: not normalised
        val = _sat((v - (wl - ww * F(0.5))) / ww)
        val = _pow(val, gamma)

        Tc = T[idx]
        Cc = C[idx]
        pos = val > 0
        a = val * ia
        alpha = _ONE - _exp(-a * step)
        if shade_on:
            # world-space gradient: index-space central difference times 0.5/voxelSize (one fp32
            # constant per axis); headlight Blinn-Phong, two-sided: n.l = |g.d| / |g|
            gx = g[0] * (F(0.5) / vsx)
            gy = g[1] * (F(0.5) / vsy)
            gz = g[2] * (F(0.5) / vsz)
            glen = np.sqrt(_dot3(gx, gy, gz, gx, gy, gz))
            ok = glen > F(e["gradEps"])
            safe = np.where(ok, glen, _ONE)
            ndl = np.fmin(np.abs(_dot3(gx, gy, gz, ddx_, ddy_, ddz_)) / safe, _ONE)
            spec = ndl
            for _ in range(int(e["specPow2"])):
                spec = spec * spec
            shade = np.where(ok, (F(e["ka"]) + F(e["kd"]) * ndl) + F(e["ks"]) * spec,
                             F(e["ka"]) + F(e["kd"]))
            emis = val * shade
            nshaded[idx] += pos.astype(np.int32)
        else:
            emis = val
        contrib = (alpha * Tc) * emis
        Cc = np.where(pos[:, None], Cc + contrib[:, None], Cc)
        Tc = np.where(pos, Tc * (_ONE - alpha), Tc)

        for show, buf, mul in ((showSeg, labels, None), (showPred, preds, F(1.5))):
            if show != 0:
                if callable(buf):      # per-sample label source (INR query / class stream), extension
                    l = np.asarray(buf(idx, nsteps[idx], qx, qy, qz)).astype(np.int64)
                else:
                    l = _sample_label(buf, qx, qy, qz, X, Y, Z).astype(np.int64)
                okl = (l > 0) & (l < 8)
                col = lut[np.where(okl, l, 0)]
                arg = -col[:, 3] * step
                if mul is not None:
                    arg = arg * mul
                al = _ONE - _exp(arg)
                at = al * Tc
                Cn = Cc + at[:, None] * col[:, :3]
                Cc = np.where(okl[:, None], Cn, Cc)
                Tc = np.where(okl, Tc * (_ONE - al), Tc)

        C[idx] = Cc
        T[idx] = Tc
        t[idx] = tt + step
        nsteps[idx] += 1

    out = np.empty((r1 - r0, Wd, 4), dtype=np.float32)
    out[..., :3] = C.reshape(r1 - r0, Wd, 3)
    out[..., 3] = _ONE
    if return_aux:
        aux = dict(nsteps=nsteps.reshape(r1 - r0, Wd), nshaded=nshaded.reshape(r1 - r0, Wd),
                   T=T.reshape(r1 - r0, Wd), fragile=fragile.reshape(r1 - r0, Wd),
                   live_samples=int(nsteps.sum()), shaded_samples=int(nshaded.sum()))
        return out, aux
    return out


# ----------------------------------------------------------------------------------------
# K2: volume_cs
# ----------------------------------------------------------------------------------------
def pack_u8_volume(voxels_u8: np.ndarray) -> np.ndarray:
    """scripts/volumeRendering/app.py:149-153: one uint32 per u8 voxel, rows of 4 (uint4)."""
    arr = np.asarray(voxels_u8, dtype=np.uint8).reshape(-1)
    pad = (-arr.size) % 4
    if pad:
        arr = np.pad(arr, (0, pad), mode="constant")
    return arr.astype(np.uint32).reshape(-1, 4)


def _fetch_k2(vol, idx, mode):
    if mode == "u32x4":          # reference layout: StructuredBuffer<uint4>, low byte used
        flat = vol.reshape(-1)
        return (flat[idx] & np.uint32(0xFF)).astype(np.float32) / F(255.0)
    if mode == "u8":
        return vol.reshape(-1)[idx].astype(np.float32) / F(255.0)
    return vol.reshape(-1)[idx]  # "f32": build-defined generalisation


def volume_cs(params: Dict[str, Any], vol: np.ndarray, *, mode: str = "u32x4",
              ext: Optional[Dict[str, Any]] = None, return_aux: bool = False,
              rows: Optional[Tuple[int, int]] = None):
    """K2: scripts/volumeRendering/volume_render.slang:104-148.  float32 (H,W,4)."""
    e = dict(DEFAULT_EXT)
    if ext:
        e.update(ext)
    Wd, Hd = int(params["imageSize"][0]), int(params["imageSize"][1])
    dx_, dy_, dz_ = (int(v) for v in params["volDim"])
    px = np.arange(Wd, dtype=np.float32)[None, :]
    py = np.arange(Hd, dtype=np.float32)[:, None]
    invx, invy = _ONE / F(Wd), _ONE / F(Hd)
    uvx = (px + F(0.5)) * invx
    uvy = (py + F(0.5)) * invy
    ndcx = uvx * F(2.0) - _ONE
    ndcy = _ONE - uvy * F(2.0)
    ex, ey, ez = _vec3(params["eye"])
    Ux, Uy, Uz = _vec3(params["U"])
    Vx, Vy, Vz = _vec3(params["V"])
    Wx, Wy, Wz = _vec3(params["W"])
    n_ = max(_ZERO, F(params["nearPlane"]))
    f_ = max(n_, F(params["farPlane"]))
    aspect = F(Wd) / max(_ONE, F(Hd))
    if int(e["cameraMode"]) == 0:
        th = _tan(F(0.5) * F(params["fovY"]))
        vx = np.broadcast_to(ndcx * aspect * th, (Hd, Wd)).astype(np.float32)
        vy = np.broadcast_to(ndcy * th, (Hd, Wd)).astype(np.float32)
        vz = _ONE

        def plane(d):
            return (((ex + Ux * (vx * d)) + Vx * (vy * d)) + Wx * (vz * d),
                    ((ey + Uy * (vx * d)) + Vy * (vy * d)) + Wy * (vz * d),
                    ((ez + Uz * (vx * d)) + Vz * (vy * d)) + Wz * (vz * d))
    else:  # orthographic extension: parallel segments between the near and far planes
        h = F(e["orthoHalfHeight"])
        sx = np.broadcast_to(ndcx * aspect * h, (Hd, Wd)).astype(np.float32)
        sy = np.broadcast_to(ndcy * h, (Hd, Wd)).astype(np.float32)

        def plane(d):
            return (((ex + Ux * sx) + Vx * sy) + Wx * d,
                    ((ey + Uy * sx) + Vy * sy) + Wy * d,
                    ((ez + Uz * sx) + Vz * sy) + Wz * d)
    nx, ny, nz = plane(n_)
    fx_, fy_, fz_ = plane(f_)
    steps = max(_ONE, F(params["stepCount"]))
    svx, svy, svz = (fx_ - nx) / steps, (fy_ - ny) / steps, (fz_ - nz) / steps
    r0, r1 = (0, Hd) if rows is None else rows
    sl = slice(r0, r1)
    posx, posy, posz, svx, svy, svz = (np.array(a[sl], dtype=np.float32).reshape(-1)
                                       for a in (nx, ny, nz, svx, svy, svz))
    n = posx.size
    accum = np.zeros(n, dtype=np.float32)
    alive = np.ones(n, dtype=bool)
    nfetch = np.zeros(n, dtype=np.int32)
    scale = F(4.0) / steps
    for _ in range(int(np.uint32(steps))):
        idx = np.nonzero(alive)[0]
        if not idx.size:
            break
        x, y, z = posx[idx], posy[idx], posz[idx]
        acc = accum[idx]
        inside = ((x < _ONE) & (y < _ONE) & (z < _ONE) & (x > -_ONE) & (y > -_ONE) & (z > -_ONE))
        do = inside & (acc < _ONE)
        if do.any():
            j = idx[do]
            u = F(0.5) * (x[do] + _ONE)
            v = F(0.5) * (y[do] + _ONE)
            w = F(0.5) * (z[do] + _ONE)
            xx = _sat(u) * (F(dx_) - _ONE)
            yy = _sat(v) * (F(dy_) - _ONE)
            zz = _sat(w) * (F(dz_) - _ONE)
            fx0, fy0, fz0 = np.floor(xx), np.floor(yy), np.floor(zz)
            p0x, p0y, p0z = fx0.astype(np.int64), fy0.astype(np.int64), fz0.astype(np.int64)
            p1x = np.minimum(p0x + 1, dx_ - 1)
            p1y = np.minimum(p0y + 1, dy_ - 1)
            p1z = np.minimum(p0z + 1, dz_ - 1)
            tx, ty, tz = xx - fx0, yy - fy0, zz - fz0

            def at(ax, ay, az):
                return _fetch_k2(vol, ax + ay * dx_ + az * dx_ * dy_, mode)
            c000, c100 = at(p0x, p0y, p0z), at(p1x, p0y, p0z)
            c010, c110 = at(p0x, p1y, p0z), at(p1x, p1y, p0z)
            c001, c101 = at(p0x, p0y, p1z), at(p1x, p0y, p1z)
            c011, c111 = at(p0x, p1y, p1z), at(p1x, p1y, p1z)
            c00, c01 = _lerp(c000, c100, tx), _lerp(c001, c101, tx)
            c10, c11 = _lerp(c010, c110, tx), _lerp(c011, c111, tx)
            c0, c1 = _lerp(c00, c10, ty), _lerp(c01, c11, ty)
            s = _lerp(c0, c1, tz) * scale
            a0 = accum[j]
            accum[j] = a0 + (_ONE - a0) * s
            nfetch[j] += 1
        posx[idx] = x + svx[idx]
        posy[idx] = y + svy[idx]
        posz[idx] = z + svz[idx]
        alive[idx] = ~(accum[idx] > F(0.995))
    out = np.empty((r1 - r0, Wd, 4), dtype=np.float32)
    out[..., :3] = accum.reshape(r1 - r0, Wd, 1)
    out[..., 3] = _ONE
    if return_aux:
        return out, dict(nfetch=nfetch.reshape(r1 - r0, Wd), live_samples=int(nfetch.sum()))
    return out


# ----------------------------------------------------------------------------------------
# K3: raymarch_cs (analytic SDF sphere tracer)
# ----------------------------------------------------------------------------------------
def raymarch_cs(params: Dict[str, Any], eye, U, V, Wv, width: int, height: int):
    """K3: scripts/raymarch/raymarch.slang:60-99; image size is the *texture* size."""
    (ox, oy, oz), (dx, dy, dz) = make_primary(width, height, params["fovY"], eye, U, V, Wv,
                                              k3_aspect=True)
    dx, dy, dz = dx.reshape(-1), dy.reshape(-1), dz.reshape(-1)
    n = dx.size
    t = np.zeros(n, dtype=np.float32)
    px, py, pz = (np.full(n, o, np.float32) for o in (ox, oy, oz))
    hit = np.zeros(n, dtype=bool)
    alive = np.ones(n, dtype=bool)
    thr, maxd = F(params["hitThreshold"]), F(params["maxDistance"])
    for _ in range(int(params["maxSteps"])):
        idx = np.nonzero(alive)[0]
        if not idx.size:
            break
        x = ox + t[idx] * dx[idx]
        y = oy + t[idx] * dy[idx]
        z = oz + t[idx] * dz[idx]
        px[idx], py[idx], pz[idx] = x, y, z
        d = np.sqrt(_dot3(x, y, z, x, y, z)) - F(0.6)
        h = d < thr
        hit[idx[h]] = True
        tn = t[idx] + np.minimum(np.maximum(d, F(0.01)), F(0.25))
        t[idx] = np.where(h, t[idx], tn)
        alive[idx] = ~h & ~(tn > maxd)
    nx, ny, nz = _normalize3(px, py, pz)
    at = np.arctan2(nz.astype(np.float64), nx.astype(np.float64)).astype(np.float32)
    u = at / (F(2.0) * F(3.14159265)) + F(0.5)
    v = ny * F(0.5) + F(0.5)
    ddx, ddy, ddz = _normalize3(dx, dy, dz)          # normalize(ray.d) again, :94
    tbg = F(0.5) * (ddy + _ONE)
    sky = [_lerp(F(a), F(b), tbg) for a, b in ((0.05, 0.2), (0.06, 0.25), (0.08, 0.3))]
    out = np.empty((n, 4), dtype=np.float32)
    out[:, 0] = np.where(hit, u, sky[0])
    out[:, 1] = np.where(hit, v, sky[1])
    out[:, 2] = np.where(hit, _ONE - u, sky[2])
    out[:, 3] = _ONE
    return out.reshape(height, width, 4)


# ----------------------------------------------------------------------------------------
# Cameras
# ----------------------------------------------------------------------------------------
class OrbitalCameraY:
    """scripts/raymarch/camera.py:9-114 (Y-up orbit camera), state machine included."""

    def __init__(self, target=None, radius=2.0, phi=math.pi * 0.5, theta=0.0,
                 min_radius=0.1, max_radius=100.0, min_phi=0.01, max_phi=math.pi - 0.01,
                 fovY_radians=math.radians(55.0)):
        self.target = (np.zeros(3, np.float32) if target is None
                       else np.asarray(target, dtype=np.float32).copy())
        self.radius, self.phi, self.theta = float(radius), float(phi), float(theta)
        self.min_radius, self.max_radius = float(min_radius), float(max_radius)
        self.min_phi, self.max_phi = float(min_phi), float(max_phi)
        self.fovY_radians = float(fovY_radians)

    def get_eye_position(self):
        s, c = math.sin(self.phi), math.cos(self.phi)
        return np.array([self.target[0] + self.radius * s * math.cos(self.theta),
                         self.target[1] + self.radius * c,
                         self.target[2] + self.radius * s * math.sin(self.theta)], dtype=np.float32)

    def get_basis(self):
        eye = self.get_eye_position()
        fwd = self.target - eye
        fn = float(np.linalg.norm(fwd))
        fwd = np.array([0, 0, -1], np.float32) if fn < 1e-6 else (fwd / fn).astype(np.float32)
        right = np.cross(fwd, np.array([0, 1, 0], np.float32))
        rn = float(np.linalg.norm(right))
        if rn < 1e-6:
            right = np.cross(fwd, np.array([0, 0, 1], np.float32))
            rn = float(np.linalg.norm(right))
        if rn > 0:
            right = (right / rn).astype(np.float32)
        up = np.cross(right, fwd).astype(np.float32)
        return eye.astype(np.float32), right, up, fwd

    def orbit(self, d_theta, d_phi):
        self.theta += float(d_theta)
        self.phi = max(self.min_phi, min(self.max_phi, self.phi + float(d_phi)))

    def zoom(self, factor):
        self.radius = max(self.min_radius, min(self.max_radius, self.radius * float(factor)))

    def pan(self, dx, dy, viewport_height=None):
        _, right, up, _ = self.get_basis()
        pixels = self._pan_pixels(viewport_height)
        px_to_world = 2.0 * self.radius * math.tan(max(1e-3, self.fovY_radians * 0.5)) / pixels
        self.target = (self.target - right * (float(dx) * px_to_world)
                       + up * (float(dy) * px_to_world)).astype(np.float32)

    @staticmethod
    def _pan_pixels(viewport_height):
        return 720.0  # scripts/raymarch/camera.py:100 ignores the viewport


class OrbitalCameraUp(OrbitalCameraY):
    """inr/viewer/camera.py:8-129 (arbitrary world_up)."""

    def __init__(self, *a, world_up=None, **kw):
        super().__init__(*a, **kw)
        self.world_up = (np.array([0, 1, 0], np.float32) if world_up is None
                         else np.asarray(world_up, dtype=np.float32))

    def _base_frame(self):
        wu = self.world_up
        ref = np.array([0, 0, 1], np.float32)
        if abs(float(np.dot(wu, ref))) > 0.999:
            ref = np.array([1, 0, 0], np.float32)
        r = np.cross(ref, wu)
        rn = float(np.linalg.norm(r))
        if rn < 1e-6:
            r, rn = np.array([1, 0, 0], np.float32), 1.0
        r = (r / rn).astype(np.float32)
        f = np.cross(wu, r).astype(np.float32)
        fn = float(np.linalg.norm(f))
        if fn > 0:
            f = (f / fn).astype(np.float32)
        return r, f, wu

    def get_eye_position(self):
        r, f, u = self._base_frame()
        s, c = math.sin(self.phi), math.cos(self.phi)
        d = (s * math.cos(self.theta)) * r + (s * math.sin(self.theta)) * f + c * u
        return (self.target + self.radius * d.astype(np.float32)).astype(np.float32)

    def get_basis(self):
        eye = self.get_eye_position()
        fwd = self.target - eye
        fn = float(np.linalg.norm(fwd))
        fwd = np.array([0, 0, -1], np.float32) if fn < 1e-6 else (fwd / fn).astype(np.float32)
        right = np.cross(fwd, self.world_up)
        rn = float(np.linalg.norm(right))
        if rn < 1e-6:
            right, _, _ = self._base_frame()
            rn = float(np.linalg.norm(right))
        if rn > 0:
            right = (right / rn).astype(np.float32)
        up = np.cross(right, fwd).astype(np.float32)
        if float(np.dot(up, self.world_up)) < 0.0:
            up, right = -up, -right
        return eye.astype(np.float32), right, up, fwd

    @staticmethod
    def _pan_pixels(viewport_height):
        ok = viewport_height is not None and viewport_height > 0
        return max(1.0, float(viewport_height) if ok else 720.0)


# ----------------------------------------------------------------------------------------
# Volume preparation (host side of K1/K2)
# ----------------------------------------------------------------------------------------
def normalize_volume(data: np.ndarray):
    """inr/viewer/brats_viewer.py:50-65 minus the NIfTI read.  (X,Y,Z) -> (linear, norm, dims)."""
    data = np.asarray(data, dtype=np.float32)
    vmin = float(np.percentile(data, 1.0))
    vmax = float(np.percentile(data, 99.5))
    if vmax <= vmin:
        vmax, vmin = float(np.max(data)), float(np.min(data))
    rng = max(1e-6, vmax - vmin)
    norm = np.clip((data - vmin) / rng, 0.0, 1.0).astype(np.float32)
    linear = np.ascontiguousarray(norm.transpose(2, 1, 0).reshape(-1))
    return linear, norm, np.array(norm.shape, dtype=np.uint32)


def flatten_labels(data: np.ndarray):
    """inr/viewer/brats_viewer.py:70-74."""
    labels = np.rint(np.asarray(data, dtype=np.float32)).astype(np.uint32)
    return np.ascontiguousarray(labels.transpose(2, 1, 0).reshape(-1)), np.array(labels.shape, np.uint32)


def world_scale(dims, zooms):
    """inr/viewer/brats_viewer.py:206-210,322-324 -> voxel_size, vol_min, target, radius."""
    dims = np.asarray(dims).astype(np.uint32)
    scale = np.float32(1.8 / float(max(dims)))
    voxel = (np.asarray(zooms, dtype=np.float32) * scale).astype(np.float32)
    ext = voxel * dims.astype(np.float32)
    vol_min = -0.5 * ext
    center = vol_min + 0.5 * ext
    return voxel, vol_min.astype(np.float32), center.astype(np.float32), float(np.linalg.norm(ext) * 0.8)


def mask_to_u8(data: np.ndarray, mode: str = "occupancy"):
    """scripts/volumeRendering/app.py:180-197 -> u8 (Z,Y,X) flattened."""
    data = np.asarray(data, dtype=np.float32)
    if mode == "occupancy":
        v = (data > 0.5).astype(np.uint8) * 255
    elif mode == "labels":
        v = np.zeros_like(data, dtype=np.uint8)
        v[np.isclose(data, 1.0)] = 85
        v[np.isclose(data, 2.0)] = 170
        v[np.isclose(data, 4.0)] = 255
    else:
        raise ValueError(f"Unknown mask_mode '{mode}'. Use 'occupancy' or 'labels'.")
    return np.transpose(v, (2, 1, 0)).copy(order="C").reshape(-1)


def bc4_decode(bc: bytes, W: int, H: int, D: int) -> np.ndarray:
    """scripts/volumeRendering/app.py:200-248.  Plain per-block loops (small inputs only)."""
    bw, bh = (W + 3) // 4, (H + 3) // 4
    if len(bc) != D * bw * bh * 8:
        raise RuntimeError(f"BC4 data size mismatch: {len(bc)} vs {D * bw * bh * 8}")
    raw = np.frombuffer(bc, dtype=np.uint8).reshape(D, bh, bw, 8)
    out = np.zeros((D, bh * 4, bw * 4), dtype=np.uint8)
    for z in range(D):
        for by in range(bh):
            for bx in range(bw):
                blk = raw[z, by, bx]
                r0, r1 = int(blk[0]), int(blk[1])
                bits = 0
                for k in range(6):
                    bits |= int(blk[2 + k]) << (8 * k)
                pal = [r0, r1]
                if r0 > r1:
                    pal += [((7 - i) * r0 + i * r1 + 3) // 7 for i in range(1, 7)]
                else:
                    pal += [((5 - i) * r0 + i * r1 + 2) // 5 for i in range(1, 5)] + [0, 255]
                for k in range(16):
                    out[z, by * 4 + k // 4, bx * 4 + k % 4] = pal[(bits >> (3 * k)) & 7]
    return out[:, :H, :W].reshape(-1)


# ----------------------------------------------------------------------------------------
# INR forward
# ----------------------------------------------------------------------------------------
def fourier_features(coords, k: int):
    """inr/inr/model.py:11-18."""
    coords = _f(coords)
    B, dim = coords.shape
    freqs = np.arange(1, k + 1)
    ang = coords[..., None] * freqs[None, None, :].astype(np.float32) * F(math.pi)
    return np.concatenate([np.sin(ang), np.cos(ang)], axis=-1).reshape(B, dim * 2 * k).astype(np.float32)


def build_input(coords, intensities, fourier_freqs: int):
    """inr/inr/model.py:21-23."""
    return np.concatenate([_f(coords), fourier_features(coords, fourier_freqs), _f(intensities)], axis=-1)


def apply_mlp(params, x):
    """inr/inr/model.py:43-50 (fp32)."""
    *hidden, last = params
    h = _f(x)
    for layer in hidden:
        h = np.maximum(h @ _f(layer["W"]) + _f(layer["b"]), _ZERO)
    return h @ _f(last["W"]) + _f(last["b"])


def predict_volume(params, case_data, fourier_freqs: int, chunk: int = 200000):
    """inr/inr/model.py:119-141."""
    mods = case_data["mods"]
    M, H, W, D = mods.shape
    grid = np.stack(np.meshgrid(np.arange(H), np.arange(W), np.arange(D), indexing="ij"), axis=-1).reshape(-1, 3)
    intens = mods.transpose(1, 2, 3, 0).reshape(-1, M)
    norm = (grid / np.array([H - 1, W - 1, D - 1])) * 2.0 - 1.0
    preds = []
    for i in range(0, len(grid), chunk):
        x = build_input(norm[i:i + chunk].astype(np.float32), intens[i:i + chunk], fourier_freqs)
        preds.append(np.argmax(apply_mlp(params, x), axis=-1).astype(np.int16))
    return np.concatenate(preds).reshape(H, W, D), case_data["seg"]


def zscore_modality(arr):
    """inr/viewer/brats_viewer.py:281-287."""
    arr = np.asarray(arr, dtype=np.float32)
    mask = arr != 0
    if mask.any():
        arr = (arr - arr[mask].mean()) / (arr[mask].std() + 1e-6)
    return arr


def siren_apply(params, x, w0: float = 30.0):
    """notebooks/neumors_inr.ipynb:1165-1178: sin(w0*(x@w)+b) first, sin(h@w+b) after, linear head."""
    n = len(params)
    h = _f(x)
    for i in range(n - 1):
        p = params[f"l{i}"]
        z = h @ _f(p["w"])
        h = np.sin(F(w0) * z + _f(p["b"])) if i == 0 else np.sin(z + _f(p["b"]))
    p = params[f"l{n - 1}"]
    return h @ _f(p["w"]) + _f(p["b"])


def inr_sample_inputs(vols, dims, zmu, zsigma, qx, qy, qz):
    """BUILD-DEFINED (BASELINE config 5): MLP inputs of a march sample at index-space position q.
    coords = 2*clamp(q,0,dim-1)/(dim-1) - 1 (fp64, rounded once — at a lattice point this is exactly
    predict_volume's coordinate, inr/inr/model.py:124-128); intensities = the four trilinear
    samples (sampleLinear), z-scored with the viewer's per-modality constants,
    (v - mu) / sigma (inr/viewer/brats_viewer.py:281-287)."""
    X, Y, Z = dims
    c = []
    for q, d in ((qx, X), (qy, Y), (qz, Z)):
        qc = np.minimum(np.maximum(q, _ZERO), F(d) - _ONE).astype(np.float64)
        c.append(((qc / np.float64(d - 1)) * 2.0 - 1.0).astype(np.float32))
    feats = []
    for m in range(4):
        v, _ = _sample_linear(vols[m], qx, qy, qz, X, Y, Z)
        feats.append((v - F(zmu[m])) / F(zsigma[m]))
    return np.stack(c, axis=-1), np.stack(feats, axis=-1)


def brats_main_inr(params, vols, mlp_params, fourier_freqs, zmu, zsigma, labels=None, ext=None, *,
                   class_stream=None, ray_offsets=None, return_aux=False, kind="fourier", w0=30.0, rows=None,
                   record=None):
    """K1 with the prediction overlay's label taken from an MLP query at every sample instead of
    sampleLabel(gPreds).  ``showPred`` must be set.  ``kind``: "fourier" = the
    reference's Fourier/ReLU MLP (inr/inr/model.py:11-50) on build_input(coords, intensities); "siren" = the
    notebook's SIREN (neumors_inr.ipynb:853-899,1165-1178) on x = (coords, intensities), ``mlp_params`` in its
    ``{"l0": {"w","b"}, ...}`` layout.  With ``class_stream``/``ray_offsets`` the labels are read from a
    per-ray stream (class of sample k of pixel p at class_stream[ray_offsets[p] + k]) — used to check the
    compositing pass against the GPU's own bf16 classes; with ``rows`` the offsets are those of the band's
    pixels.  ``record(idx, k, coords, feats, classes)`` is called for every batch of samples (tests)."""
    dims = tuple(int(v) for v in params["dims"])

    def source(idx, k, qx, qy, qz):
        if class_stream is not None and record is None:
            return class_stream[ray_offsets[idx] + k]
        c, f = inr_sample_inputs(vols, dims, zmu, zsigma, qx, qy, qz)
        if kind == "siren":
            logits = siren_apply(mlp_params, np.concatenate([c, f], axis=-1), w0)
        else:
            logits = apply_mlp(mlp_params, build_input(c, f, fourier_freqs))
        cls = np.argmax(logits, axis=-1)
        if record is not None:
            record(idx, k, c, f, logits)
        if class_stream is not None:
            return class_stream[ray_offsets[idx] + k]
        return cls

    return brats_main(params, vols, labels, source, ext, return_aux=return_aux, rows=rows)


def model_load(npz_path, config_override=None):
    """inr/inr/model.py:217-301 (both layouts: pickled list under 'params', or flat W_i/b_i
    as written by inr/inr/train.py:216-223)."""
    npz_path = pathlib.Path(npz_path).expanduser().resolve()
    if not npz_path.is_file():
        raise FileNotFoundError(f"no checkpoint archive at {npz_path}")
    cfg_path = npz_path.with_name(f"{npz_path.stem}_info.json")
    if not cfg_path.is_file():
        raise FileNotFoundError(f"sidecar {cfg_path.name} is missing beside {npz_path.name}")
    with np.load(str(npz_path), allow_pickle=False) as z:
        names = list(z.files)
        if all(k[:2] in ("W_", "b_") for k in names) and names:
            n = len([k for k in names if k.startswith("W_")])
            params = [{"W": z[f"W_{i}"], "b": z[f"b_{i}"]} for i in range(n)]
        else:
            raise KeyError(f"{npz_path} has no W_i/b_i arrays (entries: {names})")
    config = json.loads(cfg_path.read_text())
    if config_override is not None:
        config = {**config, **config_override}
    return params, config
