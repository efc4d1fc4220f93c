"""Headless BraTS viewer: the reference's frame loop (inr/viewer/brats_viewer.py) without the
window, written against the slangpy-shaped shim so it reads like the reference's own code.

What is mirrored (file:line of brats_viewer.py):
  * defaults of the render state (:124-144): ww/wl 1/0.5, intensityAlpha 0.4, gamma 1,
    step 0.05, bg black, LUT rows 1-4;
  * ``load_dir`` (:188-248): *.nii.gz by BraTS suffix (t1n/t1c/t2w/t2f, seg), percentile
    normalisation, x-fastest upload, world scale 1.8/max(dims), dummy buffers;
  * ``frame_volume`` (:320-324): target = box centre, radius = 0.8 * |extent|;
  * ``on_click_load_inr`` (:250-310): model_load -> z-score over non-zero voxels ->
    predict_volume -> x-fastest uint32 label buffer -> showPred;
  * ``run`` (:369-450): per frame build the gParams dict, bind buffers by name, dispatch.
The window, UI sliders and swap-chain blit are out of scope; frames come back as tensors or PNGs.
"""
from __future__ import annotations

import math
import pathlib
import struct
import zlib
from typing import Dict, Optional

import numpy as np
import torch

from . import inr as inr_model
from . import nifti, shim, volume
from .camera import OrbitalCamera

MOD_SUFFIXES = {"t1n": "T1n", "t1c": "T1c", "t2w": "T2w", "t2f": "FLAIR"}
MOD_ORDER = ("T1n", "T1c", "T2w", "FLAIR")


def save_png(path, rgba: np.ndarray) -> None:
    """(H,W,4|3) float in [0,1] or uint8 -> 8-bit RGBA PNG (zlib only; no imaging dependency)."""
    a = np.asarray(rgba)
    if a.dtype != np.uint8:
        a = (np.clip(a.astype(np.float32), 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
    if a.shape[-1] == 3:
        a = np.concatenate([a, np.full(a.shape[:2] + (1,), 255, np.uint8)], axis=-1)
    h, w = a.shape[:2]
    raw = b"".join(b"\x00" + a[y].tobytes() for y in range(h))

    def chunk(tag: bytes, body: bytes) -> bytes:
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)

    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + \
        chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")
    pathlib.Path(path).write_bytes(png)


class BraTSViewer:
    def __init__(self, case_dir: Optional[pathlib.Path] = None, up: str = "Y", layout: str = "auto", math_mode: str = "strict"):
        self.device = shim.Device(enable_debug_layers=True, layout=layout, math=math_mode)
        program = self.device.load_program("brats_rt.slang", ["brats_main"])
        self.kernel = self.device.create_compute_kernel(program)
        up_vec = {"X": [1, 0, 0], "Y": [0, 1, 0], "Z": [0, 0, 1]}[up.upper()]
        self.camera = OrbitalCamera(world_up=np.array(up_vec, dtype=np.float32))
        self.fov_deg = 45.0
        self.output_texture: Optional[shim.Texture] = None
        # render state, brats_viewer.py:124-144
        self.enabled = {k: True for k in MOD_ORDER}
        self.weights = {k: 1.0 for k in MOD_ORDER}
        self.ww, self.wl, self.intensity_alpha, self.gamma = 1.0, 0.5, 0.4, 1.0
        self.near_t, self.far_t, self.step_size = 0.0, 0.0, 0.05
        self.bg_color = np.zeros(3, dtype=np.float32)
        self.lut = np.zeros((8, 4), dtype=np.float32)
        self.lut[1], self.lut[2] = [0.0, 0.4, 1.0, 0.9], [0.0, 0.8, 0.0, 0.7]
        self.lut[3] = self.lut[4] = [1.0, 0.1, 0.1, 0.9]
        self.show_seg, self.show_pred = True, False
        self.vol_dims = self.voxel_size = self.vol_min = None
        self.buffers: Dict[str, shim.Buffer] = {}
        self.raw_volumes: Dict[str, np.ndarray] = {}
        self.seg_buffer = self.pred_buffer = None
        self._empty_float = self._create_buffer(np.zeros(1, dtype=np.float32))
        self._empty_uint = self._create_buffer(np.zeros(1, dtype=np.uint32))
        if case_dir is not None:
            self.load_dir(pathlib.Path(case_dir))

    def _create_buffer(self, linear: np.ndarray) -> shim.Buffer:
        b = self.device.create_buffer(element_count=linear.size, struct_size=4, usage=shim.BufferUsage.shader_resource)
        b.copy_from_numpy(linear)
        return b

    # -- loading ---------------------------------------------------------------------------
    def load_arrays(self, mods: Dict[str, np.ndarray], zooms=(1.0, 1.0, 1.0), seg: Optional[np.ndarray] = None) -> None:
        """The body of load_dir for in-memory (X,Y,Z) arrays keyed by 'T1n','T1c','T2w','FLAIR'."""
        if not mods:
            raise RuntimeError("none of the four modality volumes (t1 / t1ce / t2 / flair) could be loaded")
        dims = None
        self.raw_volumes, self.buffers = {}, {k: self._empty_float for k in MOD_ORDER}
        for key, data in mods.items():
            lin, norm, d = volume.normalize_intensity(data)
            if dims is None:
                dims = d
            elif not np.all(d == dims):
                raise RuntimeError(f"modality {key} has dims {tuple(int(v) for v in d)}, the others {tuple(int(v) for v in dims)}")
            self.raw_volumes[key] = norm          # the viewer keeps the NORMALISED array (brats_viewer.py:205,218,227)
            self.buffers[key] = self._create_buffer(lin)
        self.vol_dims = dims.astype(np.uint32)
        self.voxel_size, self.vol_min, _, _ = volume.world_frame(dims, zooms)
        self.seg_buffer = None
        if seg is not None:
            slin, sdims = volume.labels_to_uint(seg)
            if np.all(sdims == dims):
                self.seg_buffer = self._create_buffer(slin)
        self.pred_buffer, self.show_pred = None, False
        self.frame_volume()

    def load_dir(self, case_dir: pathlib.Path) -> None:
        mods, seg, zooms = {}, None, None
        for f in sorted(pathlib.Path(case_dir).glob("*.nii.gz")):
            name = f.name.lower()
            if name.endswith(("-seg.nii.gz", "_seg.nii.gz", "tumormask.nii.gz")):
                seg = nifti.read_nifti(f)[0]
                continue
            for suf, key in MOD_SUFFIXES.items():
                if name.endswith(f"-{suf}.nii.gz") or name.endswith(f"_{suf}.nii.gz"):
                    data, z = nifti.read_nifti(f)
                    mods[key] = data
                    zooms = z if zooms is None else zooms
        if not mods:
            raise RuntimeError("none of the four modality volumes (t1 / t1ce / t2 / flair) could be loaded")
        self.load_arrays(mods, zooms, seg)

    def frame_volume(self) -> None:
        if self.vol_dims is None:
            return
        ext = self.voxel_size * self.vol_dims.astype(np.float32)
        self.camera.target = (self.vol_min + 0.5 * ext).astype(np.float32)
        self.camera.radius = float(np.linalg.norm(ext) * 0.8)

    def load_inr(self, npz_path, config_override=None, allow_pickle: bool = False) -> None:
        """on_click_load_inr: prepass the whole volume through the MLP and show it as gPreds.  The reference's
        final checkpoints (train.py:386-389) pickle the parameter list; ``allow_pickle=True`` opts in to
        unpickling for a file the caller trusts (the periodic ``W_i``/``b_i`` checkpoints need no pickle)."""
        if allow_pickle:
            config_override = {**(config_override or {}), "ALLOW_PICKLE": True}
        params, config_raw = inr_model.model_load(npz_path, config_override)
        cfg = config_raw.get("config", config_raw)
        k = int(cfg.get("FOURIER_FREQS", cfg.get("fourier_freqs", 10)))
        if not all(m in self.raw_volumes for m in MOD_ORDER):
            raise RuntimeError("the INR prepass needs all four modalities loaded")
        mods = np.stack([volume.zscore_nonzero(self.raw_volumes[m]) for m in MOD_ORDER], axis=0)
        pred, _ = inr_model.predict_volume(params, {"mods": mods, "seg": None}, fourier_freqs=k)
        self.pred_buffer = shim.Buffer(self.device, pred.numel(), 4)
        self.pred_buffer.tensor = inr_model.labels_for_viewer(pred)
        self.show_pred = True

    # -- frame -----------------------------------------------------------------------------
    def params(self, width: int, height: int) -> dict:
        eye, gU, gV, gW = self.camera.get_basis()
        return {
            "imageSize": (width, height), "fovY": math.radians(self.fov_deg),
            "eye": eye, "U": gU, "V": gV, "W": gW,
            "volMin": self.vol_min, "voxelSize": self.voxel_size, "dims": self.vol_dims,
            "stepSize": self.step_size, "nearT": self.near_t, "farT": self.far_t, "bgColor": self.bg_color,
            "volEnabled": tuple(np.uint32(1 if self.enabled.get(k) and k in self.raw_volumes else 0) for k in MOD_ORDER),
            "volWeight": tuple(float(self.weights[k]) for k in MOD_ORDER),
            "ww": self.ww, "wl": self.wl, "intensityAlpha": self.intensity_alpha,
            "gamma": self.gamma, "gradBoost": 1.5, "gradScale": 1.0,
            "showSeg": 1 if (self.show_seg and self.seg_buffer) else 0,
            "showPred": 1 if (self.show_pred and self.pred_buffer) else 0,
            "lutColorAlpha": [tuple(map(float, row)) for row in self.lut.tolist()],
        }

    def render(self, width: int = 1280, height: int = 720, ext=None) -> shim.Texture:
        if self.vol_dims is None:
            raise RuntimeError("no volume loaded")
        if self.output_texture is None or (self.output_texture.width, self.output_texture.height) != (width, height):
            self.output_texture = self.device.create_texture(format=shim.Format.rgba16_float, width=width, height=height)
        ce = self.device.create_command_encoder()
        self.kernel.dispatch(
            thread_count=[width, height, 1],
            vars={"gOutput": self.output_texture,
                  "gIntensity0": self.buffers["T1n"], "gIntensity1": self.buffers["T1c"],
                  "gIntensity2": self.buffers["T2w"], "gIntensity3": self.buffers["FLAIR"],
                  "gLabels": self.seg_buffer or self._empty_uint, "gPreds": self.pred_buffer or self._empty_uint,
                  "gParams": self.params(width, height)},
            command_encoder=ce, ext=ext)
        self.device.submit_command_buffer(ce.finish())
        return self.output_texture

    def run(self, frames: int, width: int = 1280, height: int = 720, d_theta: float = math.radians(2.0),
            out_dir: Optional[pathlib.Path] = None) -> torch.Tensor:
        """Auto-rotating frame loop (the reference rotates by a fixed angle per frame,
        scripts/volumeRendering/app.py:376-384); returns the last frame, optionally dumping PNGs."""
        tex = None
        for i in range(frames):
            tex = self.render(width, height)
            if out_dir is not None:
                save_png(pathlib.Path(out_dir) / f"frame_{i:04d}.png", tex.to_numpy())
            self.camera.orbit(d_theta, 0.0)
        return tex.tensor
