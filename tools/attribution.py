"""Dynamic counts for the attribution of a render launch's vector instructions to the phases of the
kernel's main loop (development aid, GPU): wave-level executions of every phase per sample, from the
counters build.  Combined by hand with tools/static_blocks.py and the SQ_INSTS_VALU of the profile.
  python tools/attribution.py [scene ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd.scene import load_scene, flatten, Mesh
from raytracing_amd import lib as ht
SC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/scenes")
for name in (sys.argv[1:] or ["dragon"]):
    if name == "miata":
        from raytracing_amd.meshgen import bumpy_ellipsoid
        cam, wl = load_scene(f"{SC}/dragon.json")
        world = [wl[0], Mesh(bumpy_ellipsoid(160, 320), wl[1].triangles[0].material)]
    else:
        cam, world = load_scene(f"{SC}/{name}.json")
    W, H = (1024, 1024) if name == "cornell" else (1920, 1080)
    with ht.Scene(flatten(world)) as s:
        s.render(cam, W, H, 8)
        c = s.render(cam, W, H, 128, seed=1, collect_counters=1)["stats"]
        ns = c["samples"]
        node_w, all_w = c["n_node_steps"] / 64, c["n_steps"] / 64       # wave-level step executions
        leaf_w = all_w - node_w
        print(f"{name}: per sample: rays {c['n_rays']/ns:.3f}, box tests {c['n_box']/ns:.2f}, sphere {c['n_sphere']/ns:.2f}, "
              f"rect {c['n_rect']/ns:.2f}, triangle {c['n_tri']/ns:.2f}")
        print(f"  wave-level executions per sample: node steps {node_w/ns:.4f}, leaf steps {leaf_w/ns:.4f}, "
              f"shade rounds {c['n_shade_rounds']/ns:.4f}, regeneration rounds {c['n_regen_rounds']/ns:.4f}")
        print(f"  lanes per node/leaf step {c['n_active']/max(1,c['n_steps'])*64:.1f} of 64; "
              f"lanes shaded per shade round {(sum(c['n_shade'])+c['n_miss'])/max(1,c['n_shade_rounds']):.1f}; "
              f"camera paths started per regeneration round {ns/max(1,c['n_regen_rounds']):.1f}")
        print(f"  wave cycles per sample (counters build): traversal {c['cyc_trav']/ns:.0f}, shade {c['cyc_shade']/ns:.0f}, "
              f"claim + event + ray start {c['cyc_regen']/ns:.0f}")
