#!/usr/bin/env python3
"""Per-rank march time of config 4 (512^3, 2048^2, 512 steps, shaded, 64x64 tiles round-robin) for world sizes 1..8,
measured on ONE GPU by rendering rank 0's share: what each GPU of an N-GPU run has to do per frame (no exchange).
    python3 tools/tile_share_bench.py [variants...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mrirt
from mrirt import synth, tiles
n, image, steps, tile = 512, 2048, 512, 64
variants = [int(v) for v in sys.argv[1:]] or [0, 2]
vol = synth.synth_volume(n)
g = mrirt.upload_grid(vol, (n, n, n), "vga")
p = synth.brats_scene(n, image, steps, channels=1, intensity_alpha=16.0)
for world in (1, 2, 4, 8):
    line = f"world {world}:"
    for v in variants:
        ext = dict(synth.SHADE_EXT, layout="vga", kernelVariant=v)
        e = tiles.shard_ext(ext, 0, world, tile)
        out = mrirt.render_brats(p, [g], ext=e)
        for _ in range(3):
            mrirt.render_brats(p, [g], out=out, ext=e)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            mrirt.render_brats(p, [g], out=out, ext=e)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        _, st = mrirt.render_brats(p, [g], out=out, ext=e, stats=True)
        line += f"  variant {v}: {ms:.3f} ms ({st['live_samples'] / ms / 1e6:.1f} Gsamples/s, x{world} = {world * st['live_samples'] / ms / 1e6:.0f})"
    print(line, flush=True)
