"""Kernel time of one launch against its samples per pixel: T = a + b * spp (development aid).
  python tools/launch_cost.py [scene] [max_depth]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracing_amd.scene import load_scene, flatten
from raytracing_amd import lib as ht
SC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 50
cam, world = load_scene(f"{SC}/{name}.json")
W, H = (1024, 1024) if name == "cornell" else (1920, 1080)
with ht.Scene(flatten(world)) as s:
    s.render(cam, W, H, 8)
    xs, ys = [], []
    for spp in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512):
        t = min(s.render(cam, W, H, spp, batch=spp, seed=k, max_depth=depth)["stats"]["seconds_kernel"] for k in range(3))
        xs.append(spp); ys.append(t * 1e3)
        print(f"{name} depth {depth}: {spp:4d} spp  {t*1e3:8.3f} ms", flush=True)
    b, a = np.polyfit(xs[4:], ys[4:], 1)
    print(f"fit over 16..512 spp: a = {a:.3f} ms, b = {b:.5f} ms per spp")
