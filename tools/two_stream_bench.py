#!/usr/bin/env python3
"""Does overlapping consecutive frames hide the fill and drain of a launch?  Config 3 (512^3, 1024^2, 512 steps, shaded, VGA),
K frames back to back on ONE stream against the same K frames dealt round-robin over S streams (each frame into its own
output buffer; frames of different streams may run concurrently).  Wall-clock throughput, HIP events around the whole batch.
    python3 tools/two_stream_bench.py [streams ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mrirt
from mrirt import synth
n, image, steps, K = 512, int(os.environ.get("IMAGE", 1024)), 512, 40
vol = synth.synth_volume(n)
g = mrirt.upload_grid(vol, (n, n, n), "vga")
p = synth.brats_scene(n, image, steps, channels=1, intensity_alpha=16.0)
ext = dict(synth.SHADE_EXT, layout="vga")
ref = mrirt.render_brats(p, [g], ext=ext)
for S in [int(v) for v in sys.argv[1:]] or [1, 2, 3, 4]:
    streams = [torch.cuda.Stream() for _ in range(S)]
    outs = [torch.empty_like(ref) for _ in range(S)]
    def run(k):
        for i in range(k):
            s = streams[i % S]
            mrirt.render_brats(p, [g], out=outs[i % S], ext=ext, stream=s)
    torch.cuda.synchronize(); run(2 * S); torch.cuda.synchronize()
    t = time.perf_counter(); run(K); torch.cuda.synchronize(); dt = time.perf_counter() - t
    ok = all(torch.equal(o, ref) for o in outs)
    print(f"image {image}^2, {S} stream(s): {dt / K * 1e3:.4f} ms per frame over {K} frames (frames identical: {ok})", flush=True)
