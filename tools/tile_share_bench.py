#!/usr/bin/env python3
"""Per-rank march time of config 4 (512^3, 2048^2, 512 steps, shaded, tiles dealt round-robin) for world sizes 1..8, measured on
ONE GPU by rendering EVERY rank's share: what each GPU of an N-GPU run has to do per frame (no exchange).  Reports the slowest
and the mean rank per world size and the percentage of linear scaling the slowest rank allows.
    python3 tools/tile_share_bench.py [tile:skew[:variant] ...]      skew = a number, or 'auto' (tiles.balanced_skew: diagonal deal)
e.g. python3 tools/tile_share_bench.py 64:0 64:auto 32:auto 64:auto:8192"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mrirt
from mrirt import synth, tiles
n, image, steps = 512, 2048, 512
cases = sys.argv[1:] or ["64:0", "64:auto", "32:auto"]
vol = synth.synth_volume(n)
g = mrirt.upload_grid(vol, (n, n, n), "vga")
p = synth.brats_scene(n, image, steps, channels=1, intensity_alpha=16.0)
ext = dict(synth.SHADE_EXT, layout="vga")
for case in cases:
    tile, sk, *rest = case.split(":")
    tile = int(tile)
    ext = dict(synth.SHADE_EXT, layout="vga", kernelVariant=int(rest[0]) if rest else 0)
    base = None
    for world in (1, 2, 4, 8):
        skew = tiles.balanced_skew(image, tile, world) if sk == "auto" else int(sk)
        times, lives = [], []
        for r in range(world):
            e = tiles.shard_ext(ext, r, world, tile, skew)
            out, st = mrirt.render_brats(p, [g], ext=e, stats=True)
            for _ in range(2):
                mrirt.render_brats(p, [g], out=out, ext=e)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                mrirt.render_brats(p, [g], out=out, ext=e)
            e1.record(); torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / 10); lives.append(st["live_samples"])
            del out
        base = base or max(times)
        print(f"tile {tile:3d} skew {skew:2d} variant {ext['kernelVariant']:5d} world {world}: slowest {max(times):.3f} ms (rank {int(np.argmax(times))}), mean {np.mean(times):.3f}, fastest {min(times):.3f}; "
              f"live max/mean {max(lives) / np.mean(lives):.3f}; {100 * base / (world * max(times)):.1f} % of linear", flush=True)
