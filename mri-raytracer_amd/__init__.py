"""mrirt — MI355X-native MRI volume ray-marcher (drop-in for the reference's render call).

The importable name is ``mrirt`` (see ``mrirt.py`` at the repo root); this directory carries
the task-mandated name ``mri-raytracer_amd``.  Layout:
  csrc/       hand-written gfx950 HIP kernels + the C ABI (include/mrirt.h) -> libmrirt.so
  _lib.py     ctypes binding (fails loudly when the library is missing — no CPU fallback)
  render.py   render_brats / render_volume_u8 / render_sdf over device tensors
  shim.py     slangpy-shaped ``Device`` / ``ComputeKernel.dispatch(thread_count, vars, ...)``
  camera.py   OrbitalCamera (both reference variants)
  volume.py   load-time volume preparation (normalise, flatten, world frame, u8 pack, BC4)
  inr.py      model_load / build_input / apply_mlp / predict_volume on the MFMA MLP kernel
  tiles.py    image-tile sharding + RCCL framebuffer gather (one process per GPU)
  synth.py    deterministic synthetic scenes for tests and bench
"""
from . import _lib, camera, inr, nifti, params, render, shim, synth, tiles, torch_ops, viewer, volume  # noqa: F401
from .camera import OrbitalCamera  # noqa: F401
from .inr import apply_mlp, build_input, inr_forward, model_load, predict_volume, render_brats_inr  # noqa: F401
from .shim import Device, KernelShim  # noqa: F401
from .render import (Grid, detile, render_brats, render_sdf, render_volume_u8, tiles_for_rank,  # noqa: F401
                     unbrick_grid, upload_grid, upload_label_cells, upload_mod4)

__version__ = "0.1.0"
