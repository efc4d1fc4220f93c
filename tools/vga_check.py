import os, sys, math, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import mrirt
from mrirt import synth
dims = (45, 38, 27)
vol = synth.synth_volume(0, 1234, dims=dims)
gv, ga = mrirt.upload_grid(vol, dims, "vg"), mrirt.upload_grid(vol, dims, "vga")
ok = True
for (phi, theta, up, radius) in ((80, 25, None, 3.0), (10, 0, None, 3.0), (90, 90, None, 3.0), (90, 0, None, 3.0), (60, 200, (1, 0, 0), 2.5), (80, 25, None, 0.3)):
    cam = synth.bench_camera(radius, phi, theta, world_up=np.array(up, np.float32) if up else None)
    for shade in (True, False):
        for math_ in ("strict", "fast"):
            p = synth.brats_scene(0, 0, 160, dims=dims, image_hw=(96, 120), channels=1, intensity_alpha=16.0, camera=cam)
            ext = dict(synth.SHADE_EXT) if shade else {}
            ext.update(math=math_)
            a, sa = mrirt.render_brats(p, [gv], ext=dict(ext, layout="vg"), stats=True)
            b, sb = mrirt.render_brats(p, [ga], ext=dict(ext, layout="vga"), stats=True)
            c = mrirt.render_brats(p, [ga], ext=dict(ext, layout="vga", kernelVariant=4))      # generic kernel
            d, sd = mrirt.render_brats(p, [ga], ext=dict(ext, layout="vga", kernelVariant=64), stats=True)   # LDS-staged kernel
            same = torch.equal(a, b) and sa == sb and torch.equal(a, c) and torch.equal(a, d) and sa == sd
            ok &= same
            print(phi, theta, up, radius, shade, math_, "same" if same else f"DIFF {float((a-b).abs().max())} {float((a-c).abs().max())} slab {float((a-d).abs().max())} {sd}", sa["live_samples"])
print("ALL OK" if ok else "FAILED")
