"""ctypes binding of libmrirt.so (the C ABI declared in include/mrirt.h).

The HIP library is the product: there is NO CPU fallback.  If the shared object is missing or
stale, :func:`lib` raises — render calls never silently route anywhere else.
"""
from __future__ import annotations

import ctypes as C
import os
import pathlib
import subprocess
import sys
from typing import List, Optional

PKG_DIR = pathlib.Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
INCLUDE = PKG_DIR.parent / "include"
SO_PATH = pathlib.Path(os.environ.get("MRIRT_LIB", PKG_DIR / "libmrirt.so"))   # override: A/B builds in development
HIP_SOURCES = ["brats_march.hip", "brats_slab.hip", "brats_ring.hip", "volume_march.hip", "grid_ops.hip", "inr_mlp.hip",
               "abort_trace.cpp"]
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-Wall"]

# every extern "C" symbol include/mrirt.h declares
ABI_SYMBOLS = [
    "mrirt_render_brats", "mrirt_render_brats_ex", "mrirt_brats_sample_counts", "mrirt_brats_emit_samples",
    "mrirt_render_brats_stream", "mrirt_brats_inr_scratch_bytes", "mrirt_render_brats_inr", "mrirt_brick_elems", "mrirt_brick_grid",
    "mrirt_unbrick_grid", "mrirt_vec4_elems", "mrirt_vga_elems", "mrirt_build_vec4_grid", "mrirt_build_label_cells", "mrirt_build_mod4_grid", "mrirt_bc4_decode", "mrirt_macro_cells", "mrirt_skip_mask_words",
    "mrirt_build_macro_max", "mrirt_build_macro_labels", "mrirt_render_brats_skip", "mrirt_render_volume", "mrirt_build_cell8", "mrirt_render_sdf", "mrirt_tiles_for_rank",
    "mrirt_detile", "mrirt_inr_pack_bytes", "mrirt_inr_pack_weights", "mrirt_inr_calibrate", "mrirt_inr_forward", "mrirt_inr_forward_refined",
    "mrirt_inr_predict_volume", "mrirt_abi_version", "mrirt_status_string", "mrirt_last_hip_error",
    "mrirt_sizeof", "mrirt_brats_skip_applicable", "mrirt_brats_kernel_family", "mrirt_install_abort_trace",
]

ABI_VERSION = 4          # MRIRT_ABI_VERSION of include/mrirt.h this binding was written against
OK = 0
LAYOUT_LINEAR, LAYOUT_BRICK, LAYOUT_VG, LAYOUT_QUAD, LAYOUT_VGA, LAYOUT_LABCELL, LAYOUT_MOD4 = 0, 1, 2, 3, 4, 5, 6
MATH_STRICT, MATH_FAST = 0, 1
OUT_RGBA32F, OUT_RGBA16F = 0, 1
VOX_U32X4, VOX_U8, VOX_F32, VOX_CELL8 = 0, 1, 2, 3
INR_FOURIER_RELU, INR_SIREN = 0, 1
INR_NO_WEIGHT_STATIONARY, INR_NO_REFINE, INR_MARK_ONLY = 1, 2, 4      # MrirtInrFlags
KERNEL_NONE, KERNEL_GENERIC, KERNEL_PIPELINED, KERNEL_ROLLING, KERNEL_SLAB, KERNEL_RING = 0, 1, 2, 3, 4, 5   # MrirtKernelFamily
KERNEL_SKIPPING, KERNEL_LABEL_CELLS = 16, 32

f32, u32 = C.c_float, C.c_uint32


class BratsParams(C.Structure):
    """MrirtBratsParams == struct Params of inr/viewer/brats_rt.slang:12-31."""
    _fields_ = [
        ("imageSize", u32 * 2), ("fovY", f32), ("pad0", f32),
        ("eye", f32 * 3), ("pad1", f32),
        ("U", f32 * 3), ("pad2", f32), ("V", f32 * 3), ("pad3", f32), ("W", f32 * 3), ("pad4", f32),
        ("volMin", f32 * 3), ("pad5", f32), ("voxelSize", f32 * 3), ("pad6", f32), ("dims", u32 * 3), ("pad7", u32),
        ("stepSize", f32), ("nearT", f32), ("farT", f32), ("pad8", f32),
        ("bgColor", f32 * 3), ("pad9", f32),
        ("volEnabled", u32 * 4), ("volWeight", f32 * 4),
        ("ww", f32), ("wl", f32), ("intensityAlpha", f32), ("padInt", f32),
        ("gamma", f32), ("gradBoost", f32), ("gradScale", f32), ("padTone", f32),
        ("showSeg", u32), ("showPred", u32), ("padFlags", u32 * 2),
        ("lutColorAlpha", (f32 * 4) * 8),
    ]


class RenderExt(C.Structure):
    _fields_ = [
        ("cameraMode", u32), ("orthoHalfHeight", f32),
        ("shadeMode", u32), ("ka", f32), ("kd", f32), ("ks", f32), ("specPow2", u32), ("gradEps", f32),
        ("ertOverride", u32), ("ertThreshold", f32),
        ("math", u32), ("outFormat", u32), ("layout", u32), ("labelLayout", u32),
        ("tileSize", u32), ("tileRank", u32), ("tileWorld", u32),
        ("kernelVariant", u32), ("tileSkew", u32), ("reserved", u32),
    ]


class VolumeParams(C.Structure):
    """MrirtVolumeParams == struct Params of scripts/volumeRendering/volume_render.slang:9-21."""
    _fields_ = [
        ("imageSize", u32 * 2), ("fovY", f32), ("stepCount", f32), ("nearPlane", f32), ("farPlane", f32),
        ("eye", f32 * 3), ("padEye", f32), ("U", f32 * 3), ("padU", f32), ("V", f32 * 3), ("padV", f32),
        ("W", f32 * 3), ("padW", f32), ("volDim", u32 * 3), ("padDim", u32),
    ]


class SdfParams(C.Structure):
    """MrirtSdfParams == struct Params + gEye/gU/gV/gW of scripts/raymarch/raymarch.slang:7-21."""
    _fields_ = [
        ("imageSize", u32 * 2), ("fovY", f32), ("maxSteps", u32),
        ("maxDistance", f32), ("hitThreshold", f32), ("normalEps", f32), ("pad0", f32),
        ("gEye", f32 * 3), ("pad1", f32), ("gU", f32 * 3), ("pad2", f32), ("gV", f32 * 3), ("pad3", f32),
        ("gW", f32 * 3), ("pad4", f32),
    ]


class InrDesc(C.Structure):
    _fields_ = [
        ("kind", u32), ("numLayers", u32), ("inDim", u32), ("outDim", u32), ("hidden", u32),
        ("fourierFreqs", u32), ("numMods", u32), ("w0", f32),
        ("weights", C.c_void_p), ("biases", C.c_void_p),
        ("flags", u32), ("tieSigmas", f32),
    ]


class Skip(C.Structure):
    """MrirtSkip: macro-cell summaries + mask scratch for exact empty-space skipping."""
    _fields_ = [("macroUb", C.c_void_p * 4), ("macroSeg", C.c_void_p), ("macroPred", C.c_void_p), ("mask", C.c_void_p),
                ("maskWords", u32), ("mapReady", u32)]


class MrirtError(RuntimeError):
    def __init__(self, status: int, where: str):
        l = lib()
        msg = l.mrirt_status_string(status).decode()
        extra = f" (hipError {l.mrirt_last_hip_error()})" if status == -4 else ""
        super().__init__(f"{where}: {msg}{extra} [status {status}]")
        self.status = status


OBJ_DIR = PKG_DIR / "build"          # per-source objects (git-ignored; they do not travel to history)


def _hipcc() -> str:
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def hipcc_command(out: Optional[pathlib.Path] = None, extra: Optional[List[str]] = None) -> List[str]:
    """The one-command form of the build (every source in one hipcc call); build() compiles the same sources with the same
    flags one object at a time, in parallel, and links them."""
    return [_hipcc(), *HIPCC_FLAGS, *(extra or []), f"-I{INCLUDE}", *[str(CSRC / s) for s in HIP_SOURCES],
            "-o", str(out or SO_PATH)]


def _headers() -> List[pathlib.Path]:
    return list(CSRC.glob("*.h")) + [INCLUDE / "mrirt.h"]


def _stale() -> bool:
    if not SO_PATH.exists():
        return True
    t = SO_PATH.stat().st_mtime
    srcs = [CSRC / s for s in HIP_SOURCES] + _headers()
    return any(p.stat().st_mtime > t for p in srcs)


def build(force: bool = False, verbose: bool = False, check: bool = True) -> pathlib.Path:
    """Cross-compile every HIP source for gfx950 into the in-tree libmrirt.so (no GPU needed): one object per source, the
    stale ones in parallel, then one link.  A rebuilt library is then checked by tools/check_async_loads.py — the
    pipelined march kernels issue their gathers as inline asm and retire them with hand-counted waits, which is sound only
    as long as no instruction touches a gather destination before its wait; that scan is a correctness gate of the build,
    whichever path (``__graft_entry__.build()``, the test session's rebuild of a stale library) triggered it."""
    if not (force or _stale()):
        return SO_PATH
    from concurrent.futures import ThreadPoolExecutor
    OBJ_DIR.mkdir(exist_ok=True)
    hdr_t = max(p.stat().st_mtime for p in _headers())
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    jobs = []
    for src in HIP_SOURCES:
        sp, op = CSRC / src, OBJ_DIR / (src + ".o")
        if force or not op.exists() or op.stat().st_mtime < max(sp.stat().st_mtime, hdr_t):
            jobs.append([_hipcc(), *flags, f"-I{INCLUDE}", "-c", str(sp), "-o", str(op)])

    def run(cmd):
        return cmd, subprocess.run(cmd, capture_output=True, text=True)
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1) or 1) as pool:
        for cmd, r in pool.map(run, jobs):
            if verbose or r.returncode != 0:
                print(" ".join(cmd))
                print(r.stdout + r.stderr)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {cmd[-3]}")
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *[str(OBJ_DIR / (s + ".o")) for s in HIP_SOURCES], "-o", str(SO_PATH)]
    r = subprocess.run(link, capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(" ".join(link))
        print(r.stdout + r.stderr)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed linking libmrirt.so")
    if check:
        tool = PKG_DIR.parent / "tools" / "check_async_loads.py"
        r = subprocess.run([sys.executable, str(tool), str(SO_PATH)], capture_output=True, text=True)
        if verbose or r.returncode != 0:
            print(r.stdout + r.stderr)
        if r.returncode != 0:
            SO_PATH.unlink(missing_ok=True)         # a library that fails the gate must not be loadable
            raise RuntimeError("tools/check_async_loads.py rejected the built libmrirt.so (see output above)")
    return SO_PATH


# --- the C++ PyTorch extension (csrc/torch_binding.cpp -> libmrirt_torch.so: torch.ops.mrirt_native.*) ------------
TORCH_SO_PATH = PKG_DIR / "libmrirt_torch.so"


def torch_binding_command() -> List[str]:
    """Host-compiler command for the operator library: no device code, torch headers + libmrirt.so."""
    import torch
    from torch.utils import cpp_extension as ce
    libdir = pathlib.Path(torch.__file__).resolve().parent / "lib"
    inc = list(ce.include_paths()) + ["/opt/rocm/include", str(INCLUDE)]
    return [os.environ.get("CXX", "g++"), "-O2", "-fPIC", "-shared", "-std=c++17", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
            f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wno-deprecated-declarations",
            *[f"-I{i}" for i in inc], str(CSRC / "torch_binding.cpp"), "-o", str(TORCH_SO_PATH),
            f"-L{libdir}", "-lc10", "-ltorch_cpu", "-ltorch", "-lc10_hip", "-ltorch_hip",
            f"-L{PKG_DIR}", "-lmrirt", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{libdir}"]


def build_torch_binding(force: bool = False, verbose: bool = False) -> pathlib.Path:
    """Compile csrc/torch_binding.cpp against this interpreter's torch and the in-tree libmrirt.so."""
    src = CSRC / "torch_binding.cpp"
    stale = (not TORCH_SO_PATH.exists() or src.stat().st_mtime > TORCH_SO_PATH.stat().st_mtime
             or (INCLUDE / "mrirt.h").stat().st_mtime > TORCH_SO_PATH.stat().st_mtime)
    if force or stale:
        build()                                   # links against libmrirt.so
        cmd = torch_binding_command()
        r = subprocess.run(cmd, capture_output=True, text=True)
        if verbose or r.returncode != 0:
            print(" ".join(cmd))
            print(r.stdout + r.stderr)
        if r.returncode != 0:
            raise RuntimeError("building libmrirt_torch.so failed")
    return TORCH_SO_PATH


_LIB: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load libmrirt.so.  Raises if it is absent: the HIP library is the only render path."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not SO_PATH.exists():
        raise ImportError(
            f"{SO_PATH} is missing — the HIP extension is the product and has no fallback. "
            "Build it with `python -c \"import __graft_entry__ as g; g.build()\"`.")
    l = C.CDLL(str(SO_PATH))
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
    l.mrirt_render_brats.argtypes = [C.POINTER(BratsParams), C.POINTER(vp), vp, vp, vp, i64, vp]
    l.mrirt_render_brats_ex.argtypes = [C.POINTER(BratsParams), C.POINTER(RenderExt), C.POINTER(vp), vp, vp, vp, i64, vp, vp]
    l.mrirt_brats_sample_counts.argtypes = [C.POINTER(BratsParams), C.POINTER(RenderExt), vp, vp]
    l.mrirt_brats_emit_samples.argtypes = [C.POINTER(BratsParams), C.POINTER(RenderExt), C.POINTER(vp),
                                           C.POINTER(f32), C.POINTER(f32), vp, vp, vp, vp]
    l.mrirt_render_brats_stream.argtypes = [C.POINTER(BratsParams), C.POINTER(RenderExt), C.POINTER(vp), vp, vp, vp,
                                            vp, i64, vp, vp]
    for fn in ("mrirt_brats_sample_counts", "mrirt_brats_emit_samples", "mrirt_render_brats_stream"):
        getattr(l, fn).restype = i32
    l.mrirt_brats_inr_scratch_bytes.argtypes = [C.POINTER(BratsParams), u32]
    l.mrirt_brats_inr_scratch_bytes.restype = i64
    l.mrirt_render_brats_inr.argtypes = [C.POINTER(BratsParams), C.POINTER(RenderExt), C.POINTER(vp), vp, C.POINTER(InrDesc),
                                         C.POINTER(f32), C.POINTER(f32), u32, vp, i64, vp, i64, vp, vp]
    l.mrirt_render_brats_inr.restype = i32
    l.mrirt_brick_elems.argtypes = [C.POINTER(u32)]
    l.mrirt_brick_elems.restype = i64
    l.mrirt_brick_grid.argtypes = [vp, vp, C.POINTER(u32), u32, vp]
    l.mrirt_unbrick_grid.argtypes = [vp, vp, C.POINTER(u32), u32, vp]
    l.mrirt_vec4_elems.argtypes = [C.POINTER(u32)]
    l.mrirt_vec4_elems.restype = i64
    l.mrirt_vga_elems.argtypes = [C.POINTER(u32)]
    l.mrirt_vga_elems.restype = i64
    l.mrirt_build_vec4_grid.argtypes = [vp, vp, C.POINTER(u32), u32, vp]
    l.mrirt_build_label_cells.argtypes = [vp, vp, C.POINTER(u32), vp, vp]
    l.mrirt_build_label_cells.restype = i32
    l.mrirt_build_mod4_grid.argtypes = [C.POINTER(vp), vp, C.POINTER(u32), vp]
    l.mrirt_build_mod4_grid.restype = i32
    l.mrirt_macro_cells.argtypes = [C.POINTER(u32)]
    l.mrirt_macro_cells.restype = i64
    l.mrirt_skip_mask_words.argtypes = [C.POINTER(u32)]
    l.mrirt_skip_mask_words.restype = i64
    l.mrirt_build_macro_max.argtypes = [vp, C.POINTER(u32), vp, vp]
    l.mrirt_build_macro_max.restype = i32
    l.mrirt_build_macro_labels.argtypes = [vp, C.POINTER(u32), vp, vp]
    l.mrirt_build_macro_labels.restype = i32
    l.mrirt_render_brats_skip.argtypes = [C.POINTER(BratsParams), C.POINTER(RenderExt), C.POINTER(vp), vp, vp, C.POINTER(Skip), vp, i64, vp, vp]
    l.mrirt_render_brats_skip.restype = i32
    l.mrirt_brats_skip_applicable.argtypes = [C.POINTER(BratsParams), C.POINTER(RenderExt), C.POINTER(vp), vp, vp, C.POINTER(Skip)]
    l.mrirt_brats_skip_applicable.restype = i32
    l.mrirt_brats_kernel_family.argtypes = l.mrirt_brats_skip_applicable.argtypes
    l.mrirt_brats_kernel_family.restype = i32
    l.mrirt_install_abort_trace.argtypes = [i32]
    l.mrirt_install_abort_trace.restype = i32
    l.mrirt_build_cell8.argtypes = [vp, u32, C.POINTER(u32), vp, vp]
    l.mrirt_build_cell8.restype = i32
    l.mrirt_bc4_decode.argtypes = [vp, u32, u32, u32, vp, vp]
    l.mrirt_bc4_decode.restype = i32
    l.mrirt_render_volume.argtypes = [C.POINTER(VolumeParams), C.POINTER(RenderExt), vp, u32, vp, i64, vp, vp]
    l.mrirt_render_sdf.argtypes = [C.POINTER(SdfParams), u32, u32, vp, i64, vp]
    l.mrirt_tiles_for_rank.argtypes = [u32, u32, u32, u32, u32]
    l.mrirt_tiles_for_rank.restype = i64
    l.mrirt_detile.argtypes = [vp, vp, u32, u32, i64, u32, u32, u32, u32, vp]
    l.mrirt_inr_pack_bytes.argtypes = [C.POINTER(InrDesc)]
    l.mrirt_inr_pack_bytes.restype = i64
    l.mrirt_inr_pack_weights.argtypes = [C.POINTER(InrDesc), vp, vp, vp]
    l.mrirt_inr_forward.argtypes = [C.POINTER(InrDesc), vp, vp, i64, vp, vp, vp]
    l.mrirt_inr_forward_refined.argtypes = [C.POINTER(InrDesc), vp, vp, i64, vp, vp, vp]
    l.mrirt_inr_calibrate.argtypes = [C.POINTER(InrDesc), vp]
    l.mrirt_inr_predict_volume.argtypes = [C.POINTER(InrDesc), vp, C.POINTER(u32), vp, vp]
    l.mrirt_status_string.argtypes = [i32]
    l.mrirt_status_string.restype = C.c_char_p
    l.mrirt_sizeof.argtypes = [u32]
    l.mrirt_sizeof.restype = u32
    for fn in ("mrirt_render_brats", "mrirt_render_brats_ex", "mrirt_brick_grid", "mrirt_unbrick_grid", "mrirt_build_vec4_grid",
               "mrirt_render_volume", "mrirt_render_sdf", "mrirt_detile", "mrirt_inr_pack_weights",
               "mrirt_inr_forward", "mrirt_inr_forward_refined", "mrirt_inr_calibrate", "mrirt_inr_predict_volume",
               "mrirt_abi_version", "mrirt_last_hip_error"):
        getattr(l, fn).restype = i32
    if l.mrirt_abi_version() != ABI_VERSION:
        raise ImportError(f"ABI mismatch: {SO_PATH} reports version {l.mrirt_abi_version()}, this binding expects {ABI_VERSION} "
                          "(rebuild: python -c \"import __graft_entry__ as g; g.build()\")")
    for which, st in enumerate((BratsParams, RenderExt, VolumeParams, SdfParams, InrDesc, Skip)):
        if l.mrirt_sizeof(which) != C.sizeof(st):
            raise ImportError(f"ABI mismatch: {st.__name__} is {C.sizeof(st)} B here, {l.mrirt_sizeof(which)} B in {SO_PATH}")
    _LIB = l
    return l


def check(status: int, where: str) -> None:
    if status != OK:
        raise MrirtError(status, where)
