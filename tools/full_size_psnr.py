"""The converged-image gate at a headline frame, as a one-off measurement (minutes of CPU, so not a
test): the device (fp32, Philox, direct sampling) against the CPU oracle running the reference's own
rejection samplers (main.scala:87-93, Vec3.scala:17-34) at matched spp, PSNR on the 8-bit toRGB
codes (Vec3.scala:82-84).  The oracle renders in chunks of samples (independent seeds) so that the
run keeps printing; a second device render with another seed gives the device's own two-seed PSNR,
the noise floor of such a comparison.
  python tools/full_size_psnr.py [scene] [W] [H] [spp] [chunk]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from raytracing_amd import lib as ht
from raytracing_amd.scene import load_scene, flatten
SC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
W, H, spp, chunk = (int(x) for x in (sys.argv[2:6] + ["1920", "1080", "1024", "64"][len(sys.argv) - 2:]))
cam, world = load_scene(f"{SC}/{name}.json")
flat = flatten(world)
codes = lambda rgb: np.floor(255.0 * np.clip(np.sqrt(np.maximum(rgb, 0.0)), 0.0, 0.999))
def psnr(a, b):
    mse = ((a - b) ** 2).mean()
    return 99.0 if mse == 0 else float(10 * np.log10(255.0 ** 2 / mse))
with ht.Scene(flat) as s:
    g1 = s.render(cam, W, H, spp, seed=3)
    g2 = s.render(cam, W, H, spp, seed=4)
print(f"{name} {W}x{H} {spp} spp: device {g1['stats']['samples'] / g1['stats']['seconds_kernel'] / 1e6:.0f} Msamples/s; "
      f"device two-seed PSNR {psnr(codes(g1['rgb'].astype(np.float64)), codes(g2['rgb'].astype(np.float64))):.2f} dB", flush=True)
osc = oracle.OracleScene(flat, bvh_seed=1)
acc = np.zeros((H, W, 3))
t0 = time.time()
for k, first in enumerate(range(0, spp, chunk)):
    n = min(chunk, spp - first)
    acc += osc.render(cam, W, H, n, sample_offset=first, sampler=oracle.OR_SAMPLER_REFERENCE, seed=1000 + k)["rgb"] * n
    print(f"  oracle: {first + n} of {spp} spp, {time.time() - t0:.0f} s", flush=True)
ref = acc / spp
binned = lambda rgb: rgb[:H // 8 * 8, :W // 8 * 8].reshape(H // 8, 8, W // 8, 8, 3).mean(axis=(1, 3))
b1 = psnr(codes(binned(ref)), codes(binned(g1["rgb"].astype(np.float64))))
bf = psnr(codes(binned(g1["rgb"].astype(np.float64))), codes(binned(g2["rgb"].astype(np.float64))))
print(f"{name} {W}x{H} {spp} spp, 8x8 pixel blocks of linear radiance averaged: device vs oracle {b1:.2f} dB, device two-seed {bf:.2f} dB")
p1, p2 = psnr(codes(ref), codes(g1["rgb"].astype(np.float64))), psnr(codes(ref), codes(g2["rgb"].astype(np.float64)))
print(f"{name} {W}x{H} {spp} spp: device vs oracle (reference samplers) {min(p1, p2):.2f} dB (the two device seeds: {p1:.2f}, {p2:.2f}); "
      f"oracle {W * H * spp / (time.time() - t0) / 1e6:.1f} Msamples/s on {len(os.sched_getaffinity(0))} cores; "
      f"mean radiance device {g1['rgb'].mean():.5f}, oracle {ref.mean():.5f}")
