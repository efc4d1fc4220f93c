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
        skew = tiles.balanced_skew(w, tile, world)                 # the diagonal deal the sharded render uses
        local = torch.empty((n_local, tile, tile, 4))
        for lt in range(n_local):
            x0, y0 = tiles.tile_origin(rank + lt * world, w, tile, skew)
            local[lt] = torch.from_numpy(padded[y0:y0 + tile, x0:x0 + tile])
        frame = tiles.gather_frame(local, w, h, tile, dst=0, all_ranks=all_ranks, skew=skew)
        if all_ranks or rank == 0:
            ok = frame is not None and torch.equal(frame, torch.from_numpy(full))
        else:
            ok = frame is None
        # the frame-loop form, as bench.py --gpus N runs it: an asynchronous exchange with three slots, 8 frames in flight order
        D = 3
        ex = tiles.FrameExchange(w, h, tile, torch.float32, "cpu", depth=D, dst=0, skew=skew)
        assert ex.n_local == n_local and ex.stream(0) is None       # (host tensors have no slot streams)
        got = []
        for f in range(8):
            slot = f % D
            if f >= D:
                got.append(ex.finish(slot))
            ex.local(slot).copy_(local + float(f))
            ex.submit(slot)
        for f in range(8 - D, 8):
            got.append(ex.finish(f % D))
        for f, fr in enumerate(got):
            ok = ok and ((fr is None) if rank != 0 else torch.equal(fr, torch.from_numpy(full) + float(f)))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _run(world, target, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, *args, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    return dict(q.get(timeout=5) for _ in range(world))


@pytest.mark.parametrize("w,h,tile,all_ranks", [(96, 64, 32, False), (100, 70, 32, False), (100, 70, 32, True)])
def test_two_rank_tile_gather(w, h, tile, all_ranks):
    assert _run(2, _worker, w, h, tile, all_ranks) == {0: True, 1: True}


def test_four_ranks_with_unequal_tile_counts():
    """100 x 70 px in 32-px tiles = 4 x 3 = 12 tiles ... over 4 ranks that is 3 each; 150 x 70 = 5 x 3 = 15
    tiles: ranks 0-2 own 4, rank 3 owns 3 (padded for the collective, dropped by the de-tiling)."""
    from mrirt import tiles
    assert [tiles.local_tile_count(150, 70, 32, r, 4) for r in range(4)] == [4, 4, 4, 3]
    assert _run(4, _worker, 150, 70, 32, False) == {r: True for r in range(4)}


def _subgroup_worker(rank, world, port, q):
    """The exchange inside a SUB-group whose root is not global rank 0: group ranks [1, 3] of a world of 4,
    dst = group rank 1 = global rank 3 (dst must be translated to the global rank)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mrirt import tiles
        members = [1, 3]
        group = dist.new_group(ranks=members)            # every rank must take part in the creation
        ok = True
        if rank in members:
            w, h, tile = 100, 70, 32
            grank, gworld = members.index(rank), len(members)
            frame = torch.arange(96 * 128 * 4, dtype=torch.float32).reshape(96, 128, 4)
            n_local = tiles.local_tile_count(w, h, tile, grank, gworld)
            local = torch.empty((n_local, tile, tile, 4))
            for lt in range(n_local):
                x0, y0 = tiles.tile_origin(grank + lt * gworld, w, tile)
                local[lt] = frame[y0:y0 + tile, x0:x0 + tile]
            out = tiles.gather_frame(local, w, h, tile, group=group, dst=1)
            ok = (out is None) if grank != 1 else torch.equal(out, frame[:h, :w])
            ex = tiles.FrameExchange(w, h, tile, torch.float32, "cpu", group=group, depth=2, dst=1)
            ex.local(0).copy_(local)
            ex.submit(0)
            fr = ex.finish(0)
            ok = ok and ((fr is None) if grank != 1 else torch.equal(fr, frame[:h, :w]))
            ok = ok and (ex._recv[0] is None) == (grank != 1)          # non-roots hold no receive buffer
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_exchange_inside_a_subgroup_with_a_nonzero_root():
    assert _run(4, _subgroup_worker) == {r: True for r in range(4)}
