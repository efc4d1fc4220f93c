"""N>1 path on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.  Each rank produces its
round-robin tiles of one frame (the *oracle* stands in for the GPU renderer here — tests may use
it), then the product's exchange step (tiles.gather_frame: pad, gather / all-gather, de-tile)
must reassemble exactly the single-process frame."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, w, h, tile, all_ranks, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mrirt
        from mrirt import synth, tiles
        from oracle import oracle_c
        dims = (20, 18, 14)
        vol = synth.synth_volume(0, 1234, dims=dims)
        p = synth.brats_scene(0, 0, 48, dims=dims, image_hw=(h, w), channels=1, intensity_alpha=16.0)
        full = oracle_c.brats_main(p, [vol])                       # (h, w, 4)
        tx, ty = (w + tile - 1) // tile, (h + tile - 1) // tile
        padded = np.zeros((ty * tile, tx * tile, 4), np.float32)
        padded[..., :3] = np.asarray(p["bgColor"], np.float32)
        padded[..., 3] = 1.0
        padded[:h, :w] = full
        n_local = tiles.local_tile_count(w, h, tile, rank, world)
        local = torch.empty((n_local, tile, tile, 4))
        for lt in range(n_local):
            x0, y0 = tiles.tile_origin(rank + lt * world, w, tile)
            local[lt] = torch.from_numpy(padded[y0:y0 + tile, x0:x0 + tile])
        frame = tiles.gather_frame(local, w, h, tile, dst=0, all_ranks=all_ranks)
        if all_ranks or rank == 0:
            ok = frame is not None and torch.equal(frame, torch.from_numpy(full))
        else:
            ok = frame is None
        # the frame-loop form: double-buffered asynchronous exchange, 5 frames in flight order
        ex = tiles.FrameExchange(w, h, tile, torch.float32, "cpu", depth=2, dst=0)
        assert ex.n_local == n_local
        got = []
        for f in range(5):
            slot = f % 2
            if f >= 2:
                got.append(ex.finish(slot))
            ex.local(slot).copy_(local + float(f))
            ex.submit(slot)
        for f in range(3, 5):
            got.append(ex.finish(f % 2))
        for f, fr in enumerate(got):
            ok = ok and ((fr is None) if rank != 0 else torch.equal(fr, torch.from_numpy(full) + float(f)))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("w,h,tile,all_ranks", [(96, 64, 32, False), (100, 70, 32, False), (100, 70, 32, True)])
def test_two_rank_tile_gather(w, h, tile, all_ranks):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, w, h, tile, all_ranks, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = dict(q.get(timeout=5) for _ in range(world))
    assert got == {0: True, 1: True}
