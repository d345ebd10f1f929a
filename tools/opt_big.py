"""Hierarchy optimiser over large meshes (HBM store): build time, cost, throughput.  Development aid.
python tools/opt_big.py [nu:nv ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd.scene import load_scene, flatten, Mesh
from raytracing_amd.meshgen import bumpy_ellipsoid
from raytracing_amd import lib as ht
SC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/scenes")
cam, wl = load_scene(f"{SC}/dragon.json")
W, H, spp = 1920, 1080, 256
for spec in (sys.argv[1:] or ["160:320"]):
    nu, nv = map(int, spec.split(":"))
    flat = flatten([wl[0], Mesh(bumpy_ellipsoid(nu, nv), wl[1].triangles[0].material)])
    for maxn in (1 << 13, 1 << 22):
        ht.set_option("bvh_optimise_max_nodes", maxn)
        t0 = time.perf_counter()
        with ht.Scene(flat) as s:
            tb = time.perf_counter() - t0
            i = s.info()
            s.render(cam, W, H, 8)
            t = min(s.render(cam, W, H, spp, batch=spp, seed=k)["stats"]["seconds_kernel"] for k in range(2))
            c = s.render(cam, W, H, 32, seed=1, collect_counters=1)["stats"]
            print(f"mesh {spec} ({2*nu*nv} triangles) optimiser up to {maxn} nodes: build {tb:.2f} s  {W*H*spp/t/1e9:6.2f} G samples/s  "
                  f"sah {i['sah_before']:.1f} -> {i['sah_after']:.1f} depth {i['bvh_depth']} box/ray {c['n_box']/c['n_rays']:.2f} "
                  f"prim/ray {(c['n_sphere']+c['n_tri']+c['n_rect'])/c['n_rays']:.2f}", flush=True)
