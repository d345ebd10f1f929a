"""Quick device-vs-oracle check used while developing (not part of the test suite)."""
import json, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd.scene import load_scene, flatten
from raytracing_amd import lib as ht, _abi as A
import oracle

SC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/scenes")
names = sys.argv[1:] or ["scene", "dragon", "cornell", "random", "perlin", "checker", "light", "plane", "triangles"]
print("devices", ht.device_count(), flush=True)
for name in names:
    cam, world = load_scene(f"{SC}/{name}.json")
    flat = flatten(world)
    osc = oracle.OracleScene(flat, bvh_seed=1)
    W, H = 320, 180
    with ht.Scene(flat) as s:
        info = s.info()
        ofh = osc.first_hit(cam, W, H)
        for prec in (1, 0):
            t0 = time.time()
            fh = s.first_hit(cam, W, H, precision=prec)
            dt = time.time() - t0
            idm = (fh["prim_id"] != ofh["prim_id"])
            hit = (ofh["prim_id"] >= 0) & ~idm
            terr = np.abs(fh["t64"][hit] - ofh["t"][hit]) / np.abs(ofh["t"][hit])
            nerr = np.linalg.norm(fh["normal64"][hit] - ofh["normal"][hit], axis=-1) / np.linalg.norm(ofh["normal"][hit], axis=-1)
            print(f"{name:10s} prec={prec} nodes={info['n_nodes']} depth={info['bvh_depth']} lds={info['lds_bytes_fp32']} id_mismatch={idm.sum()} "
                  f"t_relerr_max={terr.max() if terr.size else 0:.3e} n_relerr_max={nerr.max() if nerr.size else 0:.3e} ({dt:.3f}s)", flush=True)
        spp = 8
        o = osc.render(cam, W, H, spp, sampler=oracle.OR_SAMPLER_MATCHED, seed=7)
        for prec in (1, 0):
            t0 = time.time()
            r = s.render(cam, W, H, spp, precision=prec, seed=7, collect_counters=1)
            dt = time.time() - t0
            d = np.abs(r["rgb"].astype(np.float64) - o["rgb"])
            mse = float((d ** 2).mean())
            close = float((d.max(axis=-1) < 1e-3).mean())
            codes = float((r["packed"] == o["packed"]).mean())
            st = r["stats"]
            print(f"{name:10s} render prec={prec} mean_gpu={r['rgb'].mean():.5f} mean_or={o['rgb'].mean():.5f} mse={mse:.3e} "
                  f"pix_close={close:.4f} codes_equal={codes:.4f} kernel_s={st['seconds_kernel']:.4f} "
                  f"rays/sample={st['n_rays']/st['samples']:.3f} box/ray={st['n_box']/max(1,st['n_rays']):.1f} "
                  f"util={st['n_active']/max(1,st['n_steps']):.3f} ({dt:.3f}s)", flush=True)
