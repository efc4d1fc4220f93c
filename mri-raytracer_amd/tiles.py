"""Image-tile sharding across the GPUs of one node (one process per GPU) and the framebuffer
exchange.  No reference counterpart (the reference is single-device): SURVEY.md section 8(e).

Rays are independent, nothing is reduced: the W x H image is cut into ``tile`` x ``tile`` pixel
tiles dealt round-robin (tile id mod world size) so early-terminating and empty regions spread
evenly; the volume is replicated on every GPU (512 MiB for a 512^3 fp32 grid against 288 GB);
each rank renders its tiles into a compact [n_local, tile, tile, 4] buffer and ONE collective
moves them: an RCCL gather to the root rank (``torch.distributed`` backend "nccl" is RCCL on ROCm;
each peer's 1/N of the frame crosses its own direct xGMI link to the root), followed by a
de-tiling copy kernel.  With the gloo backend (CPU tests) the same code path runs on host
tensors.
"""
from __future__ import annotations

from typing import Any, Dict, Mapping, Optional, Sequence

import torch
import torch.distributed as dist

DEFAULT_TILE = 64


def num_tiles(width: int, height: int, tile: int = DEFAULT_TILE) -> int:
    return ((width + tile - 1) // tile) * ((height + tile - 1) // tile)


def local_tile_count(width: int, height: int, tile: int, rank: int, world: int) -> int:
    """Tiles rank ``rank`` owns: ids rank, rank+world, ... (same rule as mrirt_tiles_for_rank)."""
    n = num_tiles(width, height, tile)
    return (n - rank + world - 1) // world if n > rank else 0


def tile_origin(tile_id: int, width: int, tile: int, skew: int = 0):
    """(x0, y0) of the dealt tile ``tile_id`` (MrirtRenderExt::tileSkew: row ty is rotated by skew * ty columns)."""
    tiles_x = (width + tile - 1) // tile
    ty = tile_id // tiles_x
    tx = (tile_id % tiles_x + (skew * ty) % tiles_x) % tiles_x
    return tx * tile, ty * tile


def balanced_skew(width: int, tile: int, world: int) -> int:
    """The ``tileSkew`` that deals tiles to ranks along diagonals: rank(tx, ty) = (tx + ty) mod world.  With the plain row-major
    deal rank = (tx + ty * tilesX) mod world, which for tilesX a multiple of world (2048 px / 64 px = 32 columns over 8 ranks)
    gives every rank whole tile COLUMNS — the ranks that draw the centre columns march the long rays (measured on config 4:
    slowest rank 0.593 ms against 0.434 for an eighth of the frame).  (tilesX - skew) mod world = 1 is the diagonal deal."""
    tiles_x = (width + tile - 1) // tile
    return (tiles_x - 1) % world if world > 1 else 0


def shard_ext(ext: Optional[Mapping[str, Any]], rank: int, world: int, tile: int = DEFAULT_TILE, skew: int = 0) -> Dict[str, Any]:
    """Render-extension dict that makes render_brats/render_volume_u8 produce this rank's tiles."""
    e = dict(ext or {})
    e.update(tileSize=tile, tileRank=rank, tileWorld=world, tileSkew=skew)
    return e


def assemble_frame(gathered: torch.Tensor, width: int, height: int, tile: int, world: int, skew: int = 0) -> torch.Tensor:
    """[world, max_local, tile, tile, 4] -> (H, W, 4).  Device tensors go through the HIP
    de-tiling kernel; host tensors (gloo tests, host read-back) through plain indexing."""
    if gathered.is_cuda:
        from .render import detile
        return detile(gathered.contiguous(), width, height, tile, world, skew=skew)
    tiles_x, tiles_y = (width + tile - 1) // tile, (height + tile - 1) // tile
    frame = gathered.new_empty((tiles_y * tile, tiles_x * tile, 4))
    for t in range(tiles_x * tiles_y):
        x0, y0 = tile_origin(t, width, tile, skew)
        frame[y0:y0 + tile, x0:x0 + tile] = gathered[t % world, t // world]
    return frame[:height, :width].contiguous()


def gather_frame(local_tiles: torch.Tensor, width: int, height: int, tile: int = DEFAULT_TILE,
                 group=None, dst: int = 0, all_ranks: bool = False, skew: int = 0) -> Optional[torch.Tensor]:
    """The one exchange step.  ``local_tiles`` is this rank's compact buffer; returns the full
    frame on ``dst`` (on every rank when ``all_ranks``), None elsewhere.  Ranks may own unequal
    tile counts (off by at most one): buffers are padded to rank 0's count for the collective."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return assemble_frame(local_tiles.unsqueeze(0), width, height, tile, 1, skew)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    max_local = local_tile_count(width, height, tile, 0, world)
    send = local_tiles
    if send.shape[0] != max_local:
        pad = send.new_zeros((max_local - send.shape[0],) + tuple(send.shape[1:]))
        send = torch.cat([send, pad], dim=0)
    send = send.contiguous()
    if all_ranks:
        # concatenated along dim 0 (the form every backend accepts), viewed as [world, max_local, ...]
        buf = send.new_empty((world * send.shape[0],) + tuple(send.shape[1:]))
        dist.all_gather_into_tensor(buf, send, group=group)
        return assemble_frame(buf.view((world,) + tuple(send.shape)), width, height, tile, world, skew)
    # ``dst`` is a rank OF THE GROUP (like ``rank`` above); torch's collectives take the global rank
    gdst = dist.get_global_rank(group, dst) if group is not None else dst
    if rank == dst:
        buf = send.new_empty((world,) + tuple(send.shape))
        dist.gather(send, list(buf.unbind(0)), dst=gdst, group=group)
        return assemble_frame(buf, width, height, tile, world, skew)
    dist.gather(send, None, dst=gdst, group=group)
    return None


class FrameExchange:
    """Double-buffered, asynchronous form of the exchange step for a frame loop.

    ``local(i)`` is the compact tile buffer the march kernel writes for frame slot i (a view of a
    buffer padded to rank 0's tile count); ``submit(i)`` starts the exchange of that slot (the
    collective runs on the backend's own stream and waits for the march kernel through stream
    ordering, so the NEXT frame's march overlaps it); ``finish(i)`` waits for it and returns the
    de-tiled frame (on ``dst`` only, None elsewhere).  Nothing is reduced, so one collective per
    frame is the whole communication:

    * ``dst`` = a rank of the group (default 0): a GATHER to that rank — each peer's 1/N of the frame
      crosses its own direct xGMI link to the root once (C4: 8 MiB per peer); the other
      ranks receive nothing;
    * ``dst=None``: every rank needs the frame — an all-gather (N times the bytes on the wire)."""

    def __init__(self, width: int, height: int, tile: int, dtype: torch.dtype, device, group=None,
                 depth: int = 2, dst: Optional[int] = 0, skew: int = 0):
        self.w, self.h, self.tile, self.group, self.dst, self.skew = width, height, tile, group, dst, skew
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_local = local_tile_count(width, height, tile, self.rank, self.world)
        self.max_local = local_tile_count(width, height, tile, 0, self.world)
        shape = (self.max_local, tile, tile, 4)
        self._send = [torch.zeros(shape, dtype=dtype, device=device) for _ in range(depth)]
        self.receives = dst is None or self.rank == dst
        # only a rank that ends up with the frame holds a receive buffer (N x the send buffer)
        self._recv = [torch.empty((self.world * self.max_local, tile, tile, 4), dtype=dtype, device=device)
                      if self.receives else None for _ in range(depth)]
        self._gdst = None
        if dst is not None and dist.is_initialized():
            self._gdst = dist.get_global_rank(group, dst) if group is not None else dst
        self._work = [None] * depth
        self._streams = [None] * depth

    def stream(self, i: int):
        """Slot i's own HIP stream (device buffers only; None for host tensors): a frame loop that runs ``finish(i)`` → march into
        ``local(i)`` → ``submit(i)`` under ``torch.cuda.stream(ex.stream(i))`` keeps ``depth`` frames in flight — frame k+1's march
        starts while frame k's drains and frame k's exchange overlaps both (a rank's share of a frame is one or two rounds of
        packets for the GPU, mostly fill and drain: DESIGN.md section 7)."""
        if not self._send[i].is_cuda:
            return None
        if self._streams[i] is None:
            self._streams[i] = torch.cuda.Stream(device=self._send[i].device)
        return self._streams[i]

    def local(self, i: int) -> torch.Tensor:
        return self._send[i][:self.n_local]

    def submit(self, i: int) -> None:
        if not dist.is_initialized():                 # single process, no group: nothing to exchange
            self._recv[i].copy_(self._send[i])
            return
        if self.dst is None:
            self._work[i] = dist.all_gather_into_tensor(self._recv[i], self._send[i], group=self.group, async_op=True)
        elif self.rank == self.dst:
            parts = list(self._recv[i].view(self.world, self.max_local, self.tile, self.tile, 4).unbind(0))
            self._work[i] = dist.gather(self._send[i], parts, dst=self._gdst, group=self.group, async_op=True)
        else:
            self._work[i] = dist.gather(self._send[i], None, dst=self._gdst, group=self.group, async_op=True)

    def finish(self, i: int) -> Optional[torch.Tensor]:
        if self._work[i] is not None:
            self._work[i].wait()          # orders the current stream after the collective
            self._work[i] = None
        if not self.receives:
            return None
        g = self._recv[i].view(self.world, self.max_local, self.tile, self.tile, 4)
        return assemble_frame(g, self.w, self.h, self.tile, self.world, self.skew)


def render_brats_sharded(params, intensities: Sequence, labels=None, preds=None, ext=None,
                         tile: int = DEFAULT_TILE, group=None, dst: int = 0, all_ranks: bool = False):
    """K1 across the process group: render this rank's tiles (dealt along diagonals: balanced_skew), then gather (see module
    docstring)."""
    from .render import render_brats
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        return render_brats(params, intensities, labels, preds, ext=ext)
    w, h = int(params["imageSize"][0]), int(params["imageSize"][1])
    skew = balanced_skew(w, tile, world)
    local = render_brats(params, intensities, labels, preds, ext=shard_ext(ext, rank, world, tile, skew))
    return gather_frame(local, w, h, tile, group=group, dst=dst, all_ranks=all_ranks, skew=skew)
