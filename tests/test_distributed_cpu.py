"""spp sharding across ranks + reduce(sum): world_size-2 gloo run on CPU.

The device renderer is not available here, so each rank renders its sample range
with the CPU oracle; what is under test is the sharding arithmetic and that the
reduced frame equals the unsharded one."""
import os
import socket
import sys

import numpy as np
import pytest

from raytracing_amd.distributed import shard_samples

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("samples,world", [(1024, 8), (10, 4), (3, 8), (0, 2), (4096, 8), (7, 7)])
def test_shard_samples_partition(samples, world):
    ranges = [shard_samples(samples, r, world) for r in range(world)]
    covered = []
    for first, count in ranges:
        covered.extend(range(first, first + count))
    assert covered == list(range(samples))
    counts = [c for _, c in ranges]
    assert max(counts) - min(counts) <= 1


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import oracle
    from raytracing_amd.distributed import render_sharded
    from raytracing_amd.scene import flatten, load_scene

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cam, w = load_scene(os.path.join(ROOT, "tests", "golden", "scenes", "scene.json"))
    osc = oracle.OracleScene(flatten(w))
    W, H, spp = 32, 18, 5

    def accumulate(first, count):
        if count == 0:
            return torch.zeros(H, W, 3, dtype=torch.float64)
        r = osc.render(cam, W, H, count, sample_offset=first, seed=11, nthreads=1)
        return torch.from_numpy(r["rgb"] * count)

    buf = render_sharded(accumulate, spp, rank, world,
                         lambda b: dist.reduce(b, dst=0, op=dist.ReduceOp.SUM))
    if rank == 0:
        np.save(out_path, (buf / spp).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_frame_equals_single_rank(tmp_path, oracle_mod, flat_scene):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    cam, _, flat = flat_scene("scene")
    whole = oracle_mod.OracleScene(flat).render(cam, 32, 18, 5, seed=11, nthreads=1)["rgb"]
    np.testing.assert_allclose(got, whole, rtol=1e-13, atol=1e-13)
