"""The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build: `make -C oracle asan`).  The oracle is the
checker every GPU parity test rests on; an out-of-bounds read in it (e.g. at a clamp edge) would silently become "the
expected image".  Runs K1 (4 modalities + both overlays, plain and shaded, camera inside the volume), K2 (three voxel
modes) and K3 on small scenes in a child interpreter with the sanitizer runtime preloaded."""
import os
import pathlib
import subprocess
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
import mrirt
from mrirt import synth
from oracle import oracle_c, oracle_np
dims = (13, 11, 9)
vols = [synth.synth_volume(0, 1234 + m, phase=0.3 * m, dims=dims) for m in range(4)]
lab = synth.synth_labels(0, dims=dims)
for kw, ext in ((dict(channels=4, show_seg=True, show_pred=True), None), (dict(channels=2, intensity_alpha=16.0), dict(synth.SHADE_EXT)),
                (dict(channels=1), dict(synth.SHADE_EXT, cameraMode=1, orthoHalfHeight=1.1))):
    p = synth.brats_scene(0, 0, 40, dims=dims, image_hw=(24, 28), **kw)
    a = oracle_c.brats_main(p, vols, lab, lab[::-1].copy(), ext)
    b = oracle_np.brats_main(p, vols, lab, lab[::-1].copy(), ext)
    assert np.array_equal(a, b)
    q = dict(p, eye=np.zeros(3, np.float32))                       # camera inside the box: t0 = 0, every clamp edge is hit
    assert np.array_equal(oracle_c.brats_main(q, vols, lab, lab, ext), oracle_np.brats_main(q, vols, lab, lab, ext))
u8 = np.rint(vols[0] * 255).astype(np.uint8)
pv = synth.volume_scene(0, 28, 24, dims=dims)
for mode, data in (("u32x4", oracle_np.pack_u8_volume(u8)), ("u8", u8), ("f32", vols[0])):
    assert np.array_equal(oracle_c.volume_cs(pv, data, mode=mode), oracle_np.volume_cs(pv, data, mode=mode))
sp, eye, U, V, W = synth.sdf_scene()
assert np.array_equal(oracle_c.raymarch_cs(sp, eye, U, V, W, 20, 16), oracle_np.raymarch_cs(sp, eye, U, V, W, 20, 16))
try:
    oracle_c.brats_main(dict(p, showPred=1), vols, lab, None)          # an overlay shown without its grid is refused, not read
    raise SystemExit("missing preds accepted")
except ValueError:
    pass
print("sanitized oracle ok")
"""


def test_c_oracle_is_clean_under_asan_and_ubsan():
    asan_rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan_rt or not os.path.isabs(asan_rt) or not os.path.exists(asan_rt):
        pytest.skip("no libasan in this toolchain")
    r = subprocess.run(["make", "-C", str(ROOT / "oracle"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, LD_PRELOAD=asan_rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", OMP_NUM_THREADS="2",
               MRIRT_ORACLE_LIB=str(ROOT / "oracle" / "liboracle_asan.so"), PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": str(ROOT)}], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "sanitized oracle ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
