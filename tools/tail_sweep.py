"""Kernel time of a 128-spp and a 1024-spp dragon launch against the size of the work units at the
end of the launch (option chunk_tail) and the unit size itself (chunk): development aid.
  python tools/tail_sweep.py [scene]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd.scene import load_scene, flatten
from raytracing_amd import lib as ht
SC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
cam, world = load_scene(f"{SC}/{name}.json")
W, H = (1024, 1024) if name == "cornell" else (1920, 1080)
with ht.Scene(flatten(world)) as s:
    s.render(cam, W, H, 8)
    for spp in (16, 128, 1024):
        for chunk in (0, 4, 8, 16):
            for tail in (1, 2, 4, 8):
                if chunk and tail > chunk:
                    continue
                ht.set_option("chunk", chunk)
                ht.set_option("chunk_tail", tail)
                t = min(s.render(cam, W, H, spp, batch=spp, seed=k)["stats"]["seconds_kernel"] for k in range(3 if spp < 1024 else 2))
                print(f"{name} {spp:5d} spp  chunk {chunk:2d} tail {tail}: {t*1e3:8.3f} ms  {W*H*spp/t/1e9:6.2f} G samples/s", flush=True)
