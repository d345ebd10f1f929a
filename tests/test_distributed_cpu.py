"""spp sharding across ranks + reduce(sum): world_size-2 gloo run on CPU.

The device renderer is not available here, so each rank renders its sample range
with the CPU oracle; what is under test is the sharding arithmetic and that the
reduced frame equals the unsharded one."""
import os
import socket
import sys

import numpy as np
import pytest

from raytracing_amd.distributed import shard_samples, shard_tiles

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


def _bench_loop_worker(rank, world, port, out_path):
    """bench.py's own frame loop (make_frame_step + timed_steps) with CPU tensors over gloo.
    The renderer is a stand-in that adds a known integer per (pixel, global sample index),
    so the reduced frame has a closed form and any lost, doubled or misplaced sample shows."""
    sys.path.insert(0, ROOT)
    import time

    import torch
    import torch.distributed as dist

    from raytracing_amd.distributed import FrameSlot, make_frame_step, timed_steps

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_pix, spp = 96, 11
    frames, calls = [], []
    pix = torch.arange(n_pix * 3, dtype=torch.int64)

    def slot():  # two frames in flight, as bench.py runs them
        accum = torch.zeros(n_pix * 3, dtype=torch.int64)

        def render_shard(first, count, timed):
            calls.append((first, count, timed))
            for s_ in range(first, first + count):
                accum.add_(pix * 1000 + s_ * s_ + 1)
            if rank == 1:
                time.sleep(0.03)  # a slow rank: the reported time must be the maximum over ranks

        return FrameSlot(accum, render_shard, lambda: frames.append(accum.clone()))

    step = make_frame_step(spp, rank, world, [slot(), slot()], dist)
    elapsed = timed_steps(step, 3, 1, world, dist, lambda: None, lambda x: torch.tensor([x], dtype=torch.float64))
    torch.save({"elapsed": elapsed, "frames": frames, "calls": calls}, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_bench_frame_loop_two_ranks_gloo(tmp_path):
    import torch
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "loop")
    mp.spawn(_bench_loop_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    n_pix, spp = 96, 11
    pix = torch.arange(n_pix * 3, dtype=torch.int64)
    whole = sum(pix * 1000 + s_ * s_ + 1 for s_ in range(spp))
    # rank 0 resolved once per step (1 warm-up + 3 timed), each time the complete frame
    assert len(r0["frames"]) == 4 and all(torch.equal(f, whole) for f in r0["frames"])
    assert r1["frames"] == []
    # contiguous sample ranges, the warm-up step untimed
    assert r0["calls"] == [(0, 6, False)] + [(0, 6, True)] * 3
    assert r1["calls"] == [(6, 5, False)] + [(6, 5, True)] * 3
    # one elapsed time for the job: the slow rank's, on both ranks
    assert r0["elapsed"] == r1["elapsed"] >= 3 * 0.03


def test_shard_tiles():
    assert shard_tiles(0, 1) == (0, 1)
    assert [shard_tiles(r, 4) for r in range(4)] == [(0, 4), (1, 4), (2, 4), (3, 4)]
    with pytest.raises(ValueError):
        shard_tiles(4, 4)


def _run_bench(*flags, env_extra=None):
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, env=env,
                       timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, [json.loads(ln) for ln in lines]


@pytest.mark.parametrize("shard", ["tiles", "samples"])
def test_bench_starts_its_own_ranks(shard):
    """Plain `python bench.py --gpus 2` (no WORLD_SIZE in the environment) must run TWO ranks: the
    script starts torch.distributed.run itself.  Driven here with the stand-in renderer over gloo
    (no GPU), through the same launcher branch and the same frame loop; the one JSON line says how
    many ranks the process group really had, and the frame equals its closed form."""
    r, out = _run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--shard", shard, "--selftest-standin")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(out) == 1, r.stdout
    line = out[0]
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["steps"] == 2
    assert line["shard"] == shard and line["frame_equals_closed_form"] is True


def test_bench_refuses_a_process_group_of_the_wrong_size():
    """--gpus 4 inside a 2-rank job is an error, not a silent 2-rank measurement."""
    r, out = _run_bench("--gpus", "4", "--selftest-standin", env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and not out
    assert "WORLD_SIZE=2" in r.stderr
