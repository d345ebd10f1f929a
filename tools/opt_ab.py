"""Hierarchy optimiser on / off, same process: throughput and box / primitive tests per ray.
Development aid.  python tools/opt_ab.py [scene ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd.scene import load_scene, flatten
from raytracing_amd import lib as ht
SC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/scenes")
for name in (sys.argv[1:] or ["dragon", "random"]):
    cam, world = load_scene(f"{SC}/{name}.json")
    flat = flatten(world)
    W, H, spp = 1920, 1080, 512 if name == "dragon" else 256
    for rnd in range(2):
        for passes in (0, 4):
            ht.set_option("bvh_optimise", passes)
            with ht.Scene(flat) as s:
                i = s.info()
                s.render(cam, W, H, 8)
                t = min(s.render(cam, W, H, spp, batch=spp, seed=k)["stats"]["seconds_kernel"] for k in range(2))
                c = s.render(cam, W, H, 32, seed=1, collect_counters=1)["stats"]
                print(f"{name} optimiser passes {passes}: {W*H*spp/t/1e9:6.2f} G samples/s  sah {i['sah_before']:.1f} -> {i['sah_after']:.1f} "
                      f"depth {i['bvh_depth']} box/ray {c['n_box']/c['n_rays']:.2f} prim/ray {(c['n_sphere']+c['n_tri']+c['n_rect'])/c['n_rays']:.2f} "
                      f"steps/sample {c['n_steps']/64/c['samples']:.4f} util {c['n_active']/max(1,c['n_steps']):.3f}", flush=True)
