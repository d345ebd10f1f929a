"""Throughput probe used while developing (not part of the test suite)."""
import sys, time, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd.scene import load_scene, flatten
from raytracing_amd import lib as ht, _abi as A

SC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/scenes")
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("scenes", nargs="*", default=["dragon", "random", "cornell"])
ap.add_argument("--w", type=int, default=1920); ap.add_argument("--h", type=int, default=1080)
ap.add_argument("--spp", type=int, default=64); ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--wait", type=int, nargs="*", default=[40]); ap.add_argument("--bpc", type=int, default=8)
ap.add_argument("--counters", type=int, default=1); ap.add_argument("--chunk", type=int, default=8); ap.add_argument("--threads", type=int, default=0); ap.add_argument("--nden", type=int, default=2); ap.add_argument("--prec", type=int, default=0); ap.add_argument("--hbm", type=int, default=0); ap.add_argument("--ctail", type=int, default=4); ap.add_argument("--wavefront", type=int, default=0); ap.add_argument("--maxwaves", type=int, default=0); ap.add_argument("--gpubvh", type=int, default=0); ap.add_argument("--wfpaths", type=int, default=1 << 24); ap.add_argument("--wfrefill", type=int, default=32); ap.add_argument("--xcd", type=int, default=1)
a = ap.parse_args()
ht.set_option("force_hbm_scene", a.hbm); 
try:
    ht.set_option("xcd_bands", a.xcd)
except ht.HipTraceError:  # an older build of the library (A/B runs)
    pass
ht.set_option("max_waves_per_cu", a.maxwaves); ht.set_option("gpu_bvh", a.gpubvh); ht.set_option("wavefront", a.wavefront); ht.set_option("wavefront_paths", a.wfpaths); ht.set_option("wavefront_refill", a.wfrefill); ht.set_option("chunk_tail", a.ctail); ht.set_option("max_blocks_per_cu", a.bpc); ht.set_option("chunk", a.chunk); ht.set_option("block_threads", a.threads); ht.set_option("node_den", a.nden)
for name in a.scenes:
    if name.startswith("mesh:"):  # mesh:<n_lat>x<n_lon>: the procedural stand-in for miata.stl
        from raytracing_amd.meshgen import bumpy_ellipsoid
        from raytracing_amd.scene import Mesh
        nl, no = (int(x) for x in name[5:].split("x"))
        cam, world = load_scene(f"{SC}/dragon.json")
        t0 = time.time()
        world = [world[0], Mesh(bumpy_ellipsoid(nl, no), world[1].triangles[0].material)]
        flat = flatten(world)
        print(f"{name}: {flat.n_prims} prims, flatten {time.time()-t0:.2f}s", flush=True)
    else:
        cam, world = load_scene(f"{SC}/{name}.json")
        flat = flatten(world)
    t0 = time.time()
    W, H = (1024, 1024) if name == "cornell" and a.w == 1920 else (a.w, a.h)
    with ht.Scene(flat) as s:
        print(f"{name}: scene create {time.time()-t0:.2f}s info={s.info()}", flush=True)
        s.render(cam, W, H, 4, precision=a.prec)  # warm-up
        for wt in a.wait:
            ht.set_option("wait_threshold", wt)
            r = s.render(cam, W, H, a.spp, precision=a.prec, batch=a.batch, seed=1)
            st = r["stats"]
            line = f"{name:8s} {W}x{H}x{a.spp} chunk={a.chunk} nden={a.nden} wait={wt} kernel_s={st['seconds_kernel']:.4f} Msamples/s={st['samples']/st['seconds_kernel']/1e6:.1f} mean={r['rgb'].mean():.5f} cfg={st['block_threads']}x{st['blocks_per_cu']} lds={st['lds_bytes']} image={s.info()['lds_bytes_fp32']} depth={s.info()['bvh_depth']}"
            if a.counters:
                r2 = s.render(cam, W, H, a.spp, precision=a.prec, batch=a.batch, seed=1, collect_counters=1)
                c = r2["stats"]; ns = c["samples"]
                flops = 40 * ns + 18 * c["n_box"] + 25 * c["n_sphere"] + 12 * c["n_rect"] + 38 * c["n_tri"] + 40 * c["n_shade"][0] + 50 * c["n_shade"][1] + 65 * c["n_shade"][2] + 700 * c["n_tex"][2] + 8 * c["n_tex"][3]
                line += f" rays/s={c['n_rays']/ns:.2f} box/ray={c['n_box']/c['n_rays']:.1f} prim/ray={(c['n_sphere']+c['n_rect']+c['n_tri'])/c['n_rays']:.2f} flop/sample={flops/ns:.0f} util={c['n_active']/max(1,c['n_steps']):.3f} iters/sample={c['n_steps']/64/ns:.3f} shade_rounds/sample={c['n_shade_rounds']/ns:.3f} lanes/shade_round={(sum(c['n_shade'])+c['n_miss'])/max(1,c['n_shade_rounds']):.1f} regen_rounds/sample={c['n_regen_rounds']/ns:.3f} lanes/regen={ns/max(1,c['n_regen_rounds']):.1f} cyc/sample trav={c['cyc_trav']/ns:.1f} shade={c['cyc_shade']/ns:.1f} regen={c['cyc_regen']/ns:.1f} clk={c['shader_clock_mhz']:.0f}MHz ctr_kernel_s={c['seconds_kernel']:.4f} TFLOPs={flops/ns*st['samples']/st['seconds_kernel']/1e12:.2f}"
            print(line, flush=True)
