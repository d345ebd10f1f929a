"""Sharding of one frame over the GPUs of a node.

Every camera path is independent (the reference's pixel value is a plain sum
over samples of one pixel, jvm/src/main/scala/main.scala:79-91), so a frame can
be cut two ways:

"samples": the samples-per-pixel range is split over ranks, each rank
  accumulates radiance SUMS for all pixels into its own buffer, and one
  reduce(sum) to rank 0 (RCCL over xGMI when the backend is "nccl") combines
  them: 24 bytes per pixel through the collective.
"tiles": the frame's 8x8-pixel tiles are dealt out round-robin (rank g owns the
  tiles g, g + G, ...), each rank renders every sample of its own tiles,
  resolves them itself, and rank 0 gathers the packed pictures: 4 bytes per
  pixel, 1/G of it from each rank, and nothing is added.

Because the random stream of a path is keyed on (pixel, GLOBAL sample index)
and the sums are integers, both give the one-rank frame bit for bit.
"""
from __future__ import annotations

from typing import Callable, List, Tuple


def shard_samples(samples: int, rank: int, world: int) -> Tuple[int, int]:
    """(first global sample index, sample count) of `rank` out of `world` ranks.

    Ranks get contiguous ranges; the first `samples % world` ranks get one more.
    """
    if samples < 0 or world < 1 or not (0 <= rank < world):
        raise ValueError("bad shard request")
    base, extra = divmod(samples, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def shard_tiles(rank: int, world: int) -> Tuple[int, int]:
    """(tile_first, tile_stride) of `rank` out of `world` ranks: the tiles rank, rank + world, ...
    of the frame's row-major tile numbering (ht_render_params.tile_first / tile_stride)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad shard request")
    return (rank, world) if world > 1 else (0, 1)


def render_sharded(accumulate: Callable[[int, int], "object"], samples: int, rank: int, world: int,
                   reduce_sum_to_root: Callable[["object"], None]):
    """Run one frame: `accumulate(first, count)` returns this rank's sum buffer
    (or None when count == 0 and the caller pre-zeroed it), then
    `reduce_sum_to_root(buffer)` adds all ranks' buffers into rank 0's."""
    first, count = shard_samples(samples, rank, world)
    buf = accumulate(first, count)
    if world > 1:
        reduce_sum_to_root(buf)
    return buf


class FrameSlot:
    """The buffers of one frame in flight.  `accum` is the tensor of int64 radiance sums
    (32.32 fixed point) on whatever device the backend of `dist` reduces on;
    `render_shard(first, count, timed)` adds this rank's sample range into it; `resolve()`
    turns the reduced sums into the picture (rank 0 only); `stream()` returns a context
    manager that makes the slot's stream current (a HIP stream per slot lets the tail of one
    frame overlap the start of the next; the default does nothing)."""

    def __init__(self, accum, render_shard: Callable[[int, int, bool], None], resolve: Callable[[], None],
                 stream: Callable[[], "object"] = None, tiles=None, tiles_all=None,
                 resolve_tiles: Callable[[], None] = None, assemble: Callable[[], None] = None):
        import contextlib

        self.accum, self.render_shard, self.resolve = accum, render_shard, resolve
        self.stream = stream or contextlib.nullcontext
        # tile sharding: `resolve_tiles()` writes this rank's own tiles into `tiles` (its packed tile
        # buffer); `tiles_all` (rank 0) receives every rank's buffer, rank g's at g * len(tiles);
        # `assemble()` (rank 0) makes the picture from it
        self.tiles, self.tiles_all, self.resolve_tiles, self.assemble = tiles, tiles_all, resolve_tiles, assemble


def make_frame_step(samples: int, rank: int, world: int, slots: List[FrameSlot], dist=None,
                    mode: str = "samples", collectives_at_one: bool = False) -> Callable[[bool], None]:
    """One frame as the benchmark and a multi-process host run it: zero this rank's sum
    buffer, add the rank's sample range into it, `dist.reduce(SUM)` to rank 0, and rank 0
    resolves the picture.  Successive frames take the slots in turn: with two slots on two
    streams the drain of frame k (its last long paths, its reduce and resolve) overlaps the
    render of frame k + 1.  `dist` is torch.distributed (None or world 1: no reduce, unless
    `collectives_at_one` asks for the gather / reduce of a one-rank group as well: a rehearsal of
    the collective calls on a machine with one GPU)."""
    if mode not in ("samples", "tiles"):
        raise ValueError("mode must be 'samples' or 'tiles'")
    first, count = shard_samples(samples, rank, world)
    frame_no = [0]
    exchange = world > 1 or (collectives_at_one and dist is not None)

    def step_tiles(timed: bool) -> None:
        # every sample of this rank's own tiles (render_shard knows which they are); the rank
        # resolves them itself and rank 0 gathers the packed tile buffers: no sums travel
        slot = slots[frame_no[0] % len(slots)]
        frame_no[0] += 1
        with slot.stream():
            slot.accum.zero_()
            if samples > 0:
                slot.render_shard(0, samples, timed)
            slot.resolve_tiles()
            if exchange:
                n = slot.tiles.numel()
                parts = [slot.tiles_all[g * n:(g + 1) * n] for g in range(world)] if rank == 0 else None
                dist.gather(slot.tiles, gather_list=parts, dst=0)
            if rank == 0:
                slot.assemble()

    if mode == "tiles":
        return step_tiles

    def step(timed: bool) -> None:
        slot = slots[frame_no[0] % len(slots)]
        frame_no[0] += 1
        with slot.stream():
            slot.accum.zero_()
            if count > 0:
                slot.render_shard(first, count, timed)
            if exchange:
                dist.reduce(slot.accum, dst=0, op=dist.ReduceOp.SUM)
            if rank == 0:
                slot.resolve()

    return step


def timed_steps(step: Callable[[bool], None], steps: int, warmup: int, world: int, dist=None,
                device_sync: Callable[[], None] = lambda: None, max_tensor: Callable[[float], "object"] = None,
                collectives_at_one: bool = False) -> float:
    """`warmup` untimed steps, then exactly `steps` timed ones bracketed by device sync +
    barrier + device sync on both sides; returns the elapsed seconds of the slowest rank
    (all_reduce MAX), the same number on every rank.  `max_tensor(x)` makes the one-element
    float64 tensor the backend can reduce (a device tensor for RCCL)."""
    import time

    exchange = world > 1 or (collectives_at_one and dist is not None)

    def fence():
        device_sync()
        if exchange:
            dist.barrier()
        device_sync()

    for _ in range(warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if exchange:
        t = max_tensor(elapsed)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed
