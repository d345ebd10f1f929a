"""How much of a launch's fixed cost comes from the expensive tiles being among the last work
units: dragon.json as it is, and with the camera turned away from the mesh (ground and sky only).
Development aid.  python tools/cold_tail.py"""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracing_amd.scene import load_scene, flatten
from raytracing_amd import lib as ht
SC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/scenes")
cam, world = load_scene(f"{SC}/dragon.json")
W, H = 1920, 1080
away = copy.deepcopy(cam)
away.target = [cam.origin[0] + (cam.origin[0] - cam.target[0]), cam.target[1], cam.origin[2] + (cam.origin[2] - cam.target[2])]
with ht.Scene(flatten(world)) as s:
    s.render(cam, W, H, 8)
    for label, c in (("towards the mesh", cam), ("away from the mesh", away)):
        for depth in (50, 4):
            xs, ys = [], []
            for spp in (16, 32, 64, 128, 256):
                t = min(s.render(c, W, H, spp, batch=spp, seed=k, max_depth=depth)["stats"]["seconds_kernel"] for k in range(3))
                xs.append(spp); ys.append(t * 1e3)
            b, a = np.polyfit(xs, ys, 1)
            print(f"{label:20s} depth {depth:2d}: a = {a:.3f} ms, b = {b:.5f} ms per spp, 128 spp {ys[3]:.3f} ms", flush=True)
