"""Sharding of one frame over the GPUs of a node.

Every camera path is independent (the reference's pixel value is a plain sum
over samples, jvm/src/main/scala/main.scala:87-91), so the samples-per-pixel
range is split over ranks, each rank accumulates radiance SUMS for all pixels
into its own buffer, and one reduce(sum) to rank 0 (RCCL over xGMI when the
backend is "nccl") combines them.  Because the random stream of a path is keyed
on the GLOBAL sample index, the set of paths is the same for every rank count.
"""
from __future__ import annotations

from typing import Callable, Tuple


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
