"""Where the time of one launch goes: work units fetched, waves that found the queue empty and
waves that ended, per 10.24 us bin (option "timeline").  Development aid.
  python tools/timeline.py [scene] [spp] [max_depth]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracing_amd.scene import load_scene, flatten
from raytracing_amd import lib as ht
SC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 128
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 50
for kv in sys.argv[4:]:  # option=value pairs for ht_set_option
    k, v = kv.split("=")
    ht.set_option(k, int(v))
cam, world = load_scene(f"{SC}/{name}.json")
W, H = (1024, 1024) if name == "cornell" else (1920, 1080)
with ht.Scene(flatten(world)) as s:
    s.render(cam, W, H, 8)
    ht.set_option("timeline", 1)
    st = s.render(cam, W, H, spp, batch=spp, seed=1, max_depth=depth)["stats"]
    ht.set_option("timeline", 0)
    tl = s.debug_timeline().astype(np.int64)
    busy = np.nonzero(tl.sum(axis=0))[0]
    # the launch is one run of bins (mod 4096): rotate so that it starts at 0
    gaps = np.diff(np.concatenate([busy, [busy[0] + 4096]]))
    start = busy[(np.argmax(gaps) + 1) % len(busy)]
    tl = np.roll(tl, -start, axis=1)
    last = np.nonzero(tl.sum(axis=0))[0].max()
    print(f"{name} {spp} spp depth {depth}: kernel {st['seconds_kernel']*1e3:.3f} ms, {tl[0].sum()} unit fetches, "
          f"{tl[1].sum()} waves found the queue empty, {tl[2].sum()} waves ended; {last + 1} bins of 10.24 us")
    step = max(1, (last + 1) // 60)
    print(" t (ms)  units fetched  queue found empty  waves ended")
    for b0 in range(0, last + 1, step):
        sl = tl[:, b0:b0 + step].sum(axis=1)
        print(f"{b0 * 0.01024:7.3f}  {sl[0]:12d}  {sl[1]:12d}  {sl[2]:12d}")
