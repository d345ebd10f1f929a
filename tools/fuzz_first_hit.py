"""Development: random small worlds, closest hit of the device (fp64 kernels, both stores, host and
device hierarchy) against the oracle.  python tools/fuzz_first_hit.py [n_worlds] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
from raytracing_amd import lib as ht, _abi as A
from raytracing_amd.scene import Camera, Diffuse, SolidColor, Sphere, Triangle, XYRect, XZRect, YZRect, flatten

n_worlds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
grey = Diffuse(SolidColor((0.7, 0.7, 0.7)))
eye = np.array((0.0, 1.0, 12.0))
bad = 0
for w in range(n_worlds):
    ns, nt, nr = (int(rng.integers(0, m)) for m in (60, 60, 8))
    world = []
    if rng.random() < 0.7:
        world.append(Sphere((0.0, -1000.5, 0.0), 1000.0, grey))
    for _ in range(ns):
        world.append(Sphere(tuple(rng.uniform(-3, 3, 3) * (1, 0.5, 1) + (0, 1.5, 0)), float(rng.uniform(0.05, 0.6)), grey))
    for _ in range(nt):
        a = rng.uniform(-3, 3, 3) * (1, 0.5, 1) + (0, 1.5, 0)
        b, c = a + rng.uniform(-0.8, 0.8, 3), a + rng.uniform(-0.8, 0.8, 3)
        if np.dot(np.cross(b - a, c - a), eye - a) < 0:
            b, c = c, b
        world.append(Triangle((tuple(a), tuple(b), tuple(c)), grey))
    for k in range(nr):
        c = rng.uniform(-2, 2, 3) * (1, 0.5, 1) + (0, 1.5, 0)
        world.append((XYRect, XZRect, YZRect)[k % 3](c[0] - 0.4, c[0] + 0.4, c[1] - 0.3, c[1] + 0.3, float(c[2]), grey))
    if not world:
        continue
    rng.shuffle(world)
    flat = flatten(world)
    cam = Camera(tuple(eye), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0), 35.0, 1.5, 0.0, -1.0, (0.5, 0.7, 1.0))
    W, H = 120, 80
    o = oracle.OracleScene(flat, bvh_seed=w).first_hit(cam, W, H)
    for hbm in (0, 1):
        for gpu in (0, 2):
            ht.set_option("force_hbm_scene", hbm); ht.set_option("gpu_bvh", gpu)
            with ht.Scene(flat, device=0) as s:
                g = s.first_hit(cam, W, H, precision=A.HT_PRECISION_FP64)
                g32 = s.first_hit(cam, W, H)
            ok = np.array_equal(g["prim_id"], o["prim_id"])
            hit = o["prim_id"] >= 0
            okt = np.allclose(g["t64"][hit], o["t"][hit], rtol=1e-9, atol=0)
            ok32 = (g32["prim_id"] == o["prim_id"]).mean()
            if not (ok and okt and ok32 > 0.99):
                bad += 1
                print(f"world {w} ({ns} spheres, {nt} triangles, {nr} rects) hbm={hbm} gpu_bvh={gpu}: ids {ok} t {okt} fp32 ids {ok32:.4f}", flush=True)
ht.set_option("force_hbm_scene", 0); ht.set_option("gpu_bvh", 0)
print(f"{n_worlds} worlds x 4 variants: {bad} mismatches")
sys.exit(1 if bad else 0)
