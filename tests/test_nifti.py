"""NIfTI-1 ingest (no GPU): round trips, gzip, byte order, scaling, and the viewer's
load_nifti_float / load_seg_uint semantics (inr/viewer/brats_viewer.py:46-74)."""
import gzip
import struct
import zlib

import numpy as np
import pytest

from mrirt import nifti, volume
from mrirt.viewer import save_png
from oracle import oracle_np as onp


def test_roundtrip_dtypes_and_gzip(tmp_path):
    rng = np.random.default_rng(0)
    for dt in (np.uint8, np.int16, np.int32, np.float32, np.float64, np.uint16):
        a = (rng.random((7, 5, 3)) * 200).astype(dt)
        for name in ("v.nii", "v.nii.gz"):
            nifti.write_nifti(tmp_path / name, a, zooms=(1.0, 0.5, 2.0))
            got, z = nifti.read_nifti(tmp_path / name)
            assert got.dtype == np.float32 and got.shape == (7, 5, 3)
            assert np.array_equal(got, a.astype(np.float32)) and np.array_equal(z, np.float32([1.0, 0.5, 2.0]))
    # x is fastest on disk
    a = np.arange(24, dtype=np.int16).reshape(4, 3, 2)
    nifti.write_nifti(tmp_path / "o.nii", a)
    raw = (tmp_path / "o.nii").read_bytes()[352:]
    assert np.array_equal(np.frombuffer(raw, "<i2")[:4], a[:, 0, 0])


def test_scaling_and_big_endian(tmp_path):
    a = np.arange(60, dtype=np.int16).reshape(5, 4, 3)
    nifti.write_nifti(tmp_path / "s.nii", a, slope=0.5, inter=-3.0)
    got, _ = nifti.read_nifti(tmp_path / "s.nii")
    assert np.array_equal(got, a.astype(np.float32) * np.float32(0.5) - np.float32(3.0))
    # hand-built big-endian file
    hdr = bytearray(352)
    struct.pack_into(">i", hdr, 0, 348)
    struct.pack_into(">8h", hdr, 40, 3, 5, 4, 3, 1, 1, 1, 1)
    struct.pack_into(">2h", hdr, 70, 4, 16)
    struct.pack_into(">8f", hdr, 76, 1.0, 2.0, 3.0, 4.0, 1.0, 1.0, 1.0, 1.0)
    struct.pack_into(">3f", hdr, 108, 352.0, 0.0, 0.0)
    hdr[344:348] = b"n+1\0"
    (tmp_path / "be.nii").write_bytes(bytes(hdr) + np.transpose(a, (2, 1, 0)).astype(">i2").tobytes())
    got, z = nifti.read_nifti(tmp_path / "be.nii")
    assert np.array_equal(got, a.astype(np.float32)) and np.array_equal(z, np.float32([2, 3, 4]))


def test_errors(tmp_path):
    (tmp_path / "short.nii").write_bytes(b"\0" * 100)
    with pytest.raises(ValueError):
        nifti.read_nifti(tmp_path / "short.nii")
    a = np.zeros((3, 3, 3), np.float32)
    nifti.write_nifti(tmp_path / "ok.nii", a)
    raw = bytearray((tmp_path / "ok.nii").read_bytes())
    raw[344:348] = b"nope"
    (tmp_path / "magic.nii").write_bytes(bytes(raw))
    with pytest.raises(ValueError, match="magic"):
        nifti.read_nifti(tmp_path / "magic.nii")
    raw = (tmp_path / "ok.nii").read_bytes()
    (tmp_path / "trunc.nii.gz").write_bytes(gzip.compress(raw[:-8]))
    with pytest.raises(ValueError, match="truncated"):
        nifti.read_nifti(tmp_path / "trunc.nii.gz")


def test_viewer_loaders_match_oracle(tmp_path):
    rng = np.random.default_rng(2)
    raw = rng.gamma(2.0, 150.0, (12, 10, 8)).astype(np.float32)
    seg = rng.integers(0, 5, (12, 10, 8)).astype(np.float32)
    nifti.write_nifti(tmp_path / "c-t1n.nii.gz", raw, zooms=(1.0, 1.0, 1.2))
    nifti.write_nifti(tmp_path / "c-seg.nii.gz", seg.astype(np.uint8))
    lin, norm, dims, zooms = nifti.load_nifti_float(tmp_path / "c-t1n.nii.gz")
    olin, onorm, odims = onp.normalize_volume(raw)
    assert np.array_equal(lin, olin) and np.array_equal(norm, onorm) and np.array_equal(dims, odims)
    assert np.array_equal(zooms, np.float32([1.0, 1.0, 1.2]))
    slin, sdims, _ = nifti.load_seg_uint(tmp_path / "c-seg.nii.gz")
    assert np.array_equal(slin, onp.flatten_labels(seg)[0]) and slin.dtype == np.uint32


def test_png_writer(tmp_path):
    img = np.zeros((5, 7, 4), np.float32)
    img[..., 0], img[..., 3] = np.linspace(0, 1, 7)[None, :], 1.0
    save_png(tmp_path / "a.png", img)
    b = (tmp_path / "a.png").read_bytes()
    assert b[:8] == b"\x89PNG\r\n\x1a\n" and b[12:16] == b"IHDR"
    w, h, depth, ctype = struct.unpack(">IIBB", b[16:26])
    assert (w, h, depth, ctype) == (7, 5, 8, 6)
    i = b.index(b"IDAT")
    n = struct.unpack(">I", b[i - 4:i])[0]
    rows = zlib.decompress(b[i + 4:i + 4 + n])
    assert len(rows) == 5 * (1 + 7 * 4) and rows[0] == 0 and rows[1 + 6 * 4] == 255
