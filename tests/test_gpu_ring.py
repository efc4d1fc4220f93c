"""The plane-synchronous LDS ring kernel (csrc/brats_ring.hip; kernelVariant bit 11): the same bits as the C oracle and as
the register-gather kernels — frame AND counters — on cameras around and inside the volume, every flat copy, both march
directions, anisotropic voxels, steps shorter and longer than a voxel; its windows must cover every read (the kernel's own
diagnostic), and it must really be the ring that serves the samples (not the per-wave gather fall-back)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RING = 2048          # kernelVariant bit 11
RING3 = 2048 | 4096  # ... with three planes in the ring instead of four (bit 12)
COUNT = 128          # bit 7: stats[1] = shaded + (reads outside a window | samples of waves that fell back)
FALLBACK = 1024      # bit 10: every wave takes the gather march


@pytest.fixture(scope="module")
def env():
    import torch
    import mrirt
    from mrirt import synth
    from oracle import oracle_c
    assert torch.cuda.is_available()
    return mrirt, synth, oracle_c


OKEYS = ("shadeMode", "ka", "kd", "ks", "specPow2", "gradEps")


@pytest.mark.parametrize("seed", range(16))
def test_ring_random_cameras_match_the_oracle(env, seed):
    mrirt, synth, oc = env
    rng = np.random.default_rng(4100 + seed)
    dims = tuple(int(v) for v in rng.integers(17, 60, 3))
    vol = synth.synth_volume(0, 20 + seed, phase=float(rng.uniform(0, 3)), dims=dims)
    ups = (None, (1.0, 0.0, 0.0), (0.0, 0.0, 1.0))
    cam = synth.bench_camera(radius=float(rng.choice([0.2, 0.9, 2.0, 3.0, 5.0])), phi_deg=float(rng.uniform(3, 177)),
                             theta_deg=float(rng.uniform(0, 360)), world_up=None if seed % 3 == 0 else np.array(ups[seed % 3], np.float32))
    shade = bool(seed & 1)
    p = synth.brats_scene(0, 0, int(rng.choice([24, 60, 150, 400])), dims=dims, image_hw=(int(rng.integers(9, 120)), int(rng.integers(9, 120))),
                          channels=1, intensity_alpha=float(rng.choice([0.4, 16.0, 60.0])), camera=cam, fov_deg=float(rng.uniform(10, 70)))
    p["voxelSize"] = (p["voxelSize"] * rng.uniform(0.6, 1.7, 3)).astype(np.float32)
    p["gamma"] = float(rng.choice([1.0, 1.0, 1.8]))
    if seed % 4 == 3:
        p["volWeight"] = (np.float32(0.7), np.float32(1), np.float32(1), np.float32(1))
    ext = dict(synth.SHADE_EXT) if shade else {}
    if seed % 5 == 4:
        ext.update(cameraMode=1, orthoHalfHeight=float(rng.uniform(0.3, 1.2)))
    ref, aux = oc.brats_main(p, [vol], None, None, {k: v for k, v in ext.items() if k in OKEYS + ("cameraMode", "orthoHalfHeight")}, return_aux=True)
    g = mrirt.upload_grid(vol, dims, "vga")
    got, st = mrirt.render_brats(p, [g], ext=dict(ext, math="strict", layout="vga", kernelVariant=RING), stats=True)
    assert np.array_equal(got.cpu().numpy(), ref), (seed, dims, float(np.abs(got.cpu().numpy() - ref).max()))
    assert st["live_samples"] == aux["live_samples"] and st["shaded_samples"] == aux["shaded_samples"]
    # three planes: lanes whose rays are out of phase wait more often, the samples they take are the same
    got3, st3 = mrirt.render_brats(p, [g], ext=dict(ext, math="strict", layout="vga", kernelVariant=RING3), stats=True)
    assert np.array_equal(got3.cpu().numpy(), ref) and st3 == st, (seed, dims)
    # tile shards (compact buffers) and the half-float target go through the same kernel
    if seed % 4 == 0:
        tiled = mrirt.render_brats(p, [g], ext=dict(ext, math="strict", layout="vga", kernelVariant=RING, tileSize=64, tileRank=0, tileWorld=1))
        plain = mrirt.render_brats(p, [g], ext=dict(ext, math="strict", layout="vga", kernelVariant=2, tileSize=64, tileRank=0, tileWorld=1))
        assert np.array_equal(tiled.cpu().numpy(), plain.cpu().numpy())
    # fast math: the ring feeds the same blend as the gather kernels
    a = mrirt.render_brats(p, [g], ext=dict(ext, math="fast", layout="vga", kernelVariant=RING))
    b = mrirt.render_brats(p, [g], ext=dict(ext, math="fast", layout="vga", kernelVariant=2))
    assert np.array_equal(a.cpu().numpy(), b.cpu().numpy())


def test_ring_windows_cover_every_read_and_the_ring_is_what_serves(env):
    """Bench-like geometry at reduced size: no read falls outside its plane window, and most samples are served from LDS."""
    mrirt, synth, oc = env
    n, image, steps = 96, 256, 192
    vol = synth.synth_volume(n, 1234)
    g = mrirt.upload_grid(vol, (n, n, n), "vga")
    for theta in (25.0, 70.0, 200.0):
        for phi in (80.0, 35.0):
            cam = synth.bench_camera(radius=3.0, phi_deg=phi, theta_deg=theta)
            p = synth.brats_scene(n, image, steps, channels=1, intensity_alpha=16.0, camera=cam)
            ext = dict(synth.SHADE_EXT, layout="vga", math="strict")
            plain, s0 = mrirt.render_brats(p, [g], ext=dict(ext, kernelVariant=2), stats=True)
            ring, s1 = mrirt.render_brats(p, [g], ext=dict(ext, kernelVariant=RING), stats=True)
            assert np.array_equal(plain.cpu().numpy(), ring.cpu().numpy())
            assert s0 == s1
            _, s2 = mrirt.render_brats(p, [g], ext=dict(ext, kernelVariant=RING | COUNT), stats=True)
            extra = s2["shaded_samples"] - s1["shaded_samples"]          # uncovered reads + samples of waves that fell back
            assert extra <= 0.25 * s1["live_samples"], (theta, phi, extra, s1)
            _, s3 = mrirt.render_brats(p, [g], ext=dict(ext, kernelVariant=RING | COUNT | FALLBACK), stats=True)
            assert s3["shaded_samples"] - s1["shaded_samples"] == s1["live_samples"]     # the hook counts what it says


def test_ring_full_size_c3_equals_the_gather_kernel(env):
    """BASELINE config 3 (512^3, 1024^2, 512 steps, shaded + ERT): the ring kernel's frame and counters are the gather
    kernel's (which the full-size property tests hold to the oracle)."""
    mrirt, synth, oc = env
    n = 512
    vol = synth.synth_volume(n, 1234)
    g = mrirt.upload_grid(vol, (n, n, n), "vga")
    del vol
    p = synth.brats_scene(n, 1024, 512, channels=1, intensity_alpha=16.0)
    ext = dict(synth.SHADE_EXT, layout="vga", math="strict")
    plain, s0 = mrirt.render_brats(p, [g], ext=dict(ext, kernelVariant=0), stats=True)
    ring, s1 = mrirt.render_brats(p, [g], ext=dict(ext, kernelVariant=RING), stats=True)
    import torch
    assert torch.equal(plain, ring)
    assert s0 == s1
    _, s2 = mrirt.render_brats(p, [g], ext=dict(ext, kernelVariant=RING | COUNT), stats=True)
    extra = s2["shaded_samples"] - s1["shaded_samples"]
    assert extra <= 0.10 * s1["live_samples"], (extra, s1)
