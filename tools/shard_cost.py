"""What one rank of an N-GPU run does, timed on ONE GPU: the launch of each rank's share of the dragon frame
(1920x1080, 1024 spp) under both sharding modes, plus this GPU's time for the steps around it.  The table
gives, per N, the slowest rank's render time; a run on N GPUs adds the exchange (reduce of the sums, or gather
of the packed tiles).  Development aid.   python tools/shard_cost.py [scene]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd.scene import load_scene, flatten
from raytracing_amd import lib as ht
from raytracing_amd.distributed import shard_samples, shard_tiles
SC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
cam, world = load_scene(f"{SC}/{name}.json")
W, H, spp = 1920, 1080, 1024
dev = torch.device("cuda", 0)
accum = torch.zeros(H * W * 3, dtype=torch.int64, device=dev)
with ht.Scene(flatten(world)) as s:
    def launch(first, count, t_first, t_stride):
        p = ht.make_params(W, H, count, sample_offset=first, seed=2025, tile_first=t_first, tile_stride=t_stride)
        best = 1e9
        for _ in range(3):
            accum.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s.render_accumulate(cam, p, accum.data_ptr(), 0, want_stats=False)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best * 1e3
    launch(0, 8, 0, 1)
    t1 = launch(0, spp, 0, 1)
    print(f"{name} 1920x1080x{spp}: one GPU {t1:.2f} ms per launch (host clock around launch + synchronize)")
    print(" N   samples mode: slowest rank ms (speed-up before the reduce)   tiles mode: slowest rank ms (speed-up before the gather)")
    for n in (2, 4, 8):
        ts = max(launch(*shard_samples(spp, r, n), 0, 1) for r in range(n))
        tt = max(launch(0, spp, *shard_tiles(r, n)) for r in range(n))
        print(f"{n:2d}   {ts:8.2f} ({t1/ts:.2f}x)                                          {tt:8.2f} ({t1/tt:.2f}x)", flush=True)
