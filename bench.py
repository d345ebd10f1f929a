#!/usr/bin/env python3
"""Headline benchmark: Msamples/s (camera paths per second) of the trace /
intersect / shade loop on scenes/dragon.json at 1920x1080, 1024 spp.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one whole frame: every rank path-traces its share of the
samples-per-pixel range for all pixels into a device buffer of radiance sums
(int64, 32.32 fixed point), the buffers are summed onto rank 0 with one reduce
(RCCL over xGMI), and rank 0
resolves the frame (mean, tone-map, row flip) on the GPU.  The frame is fixed
while N grows (strong scaling).  Rank 0 prints ONE JSON line.

Timed region: scene already resident on the device; barrier +
torch.cuda.synchronize() on both sides; max over ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene file, width, height, spp): the configurations the metric is quoted on
    "dragon": ("dragon.json", 1920, 1080, 1024),   # STL mesh + BVH, the 1/2/4/8-GPU target
    "random": ("random.json", 1920, 1080, 256),    # 484 spheres, mixed materials
    "cornell": ("cornell.json", 1024, 1024, 4096),  # emissive quad box
    "perlin": ("perlin.json", 1920, 1080, 256),
    "checker": ("checker.json", 1920, 1080, 256),
    "scene": ("scene.json", 400, 225, 16),
    # miata.json names an STL the reference does not ship: a procedural mesh of 101 760
    # triangles stands in (scene image in HBM instead of LDS)
    "miata": ("miata.json", 1920, 1080, 256),
}
FP32_VALU_PEAK_TFLOPS = 157.3  # 256 CU x 128 lanes x 2 flop x 2.4 GHz


def algorithmic_flops(c):
    """flops per the fixed counting convention: camera ray 40, slab test 18,
    sphere 25, rect 12, triangle 38, scatter 40/50/65 (Diffuse/Glossy/Glass),
    Perlin +700, Image +8; compares / selects / integer work count 0."""
    return (40 * c["samples"] + 18 * c["n_box"] + 25 * c["n_sphere"] + 12 * c["n_rect"] + 38 * c["n_tri"]
            + 40 * c["n_shade"][0] + 50 * c["n_shade"][1] + 65 * c["n_shade"][2]
            + 700 * c["n_tex"][2] + 8 * c["n_tex"][3])


def cpu_baseline(flat, cam, W, H, seconds_target=12.0):
    """The fp64 C oracle (a port of the reference's Scala loop, reference-style
    BVH and rejection samplers) on this machine's host cores, on a bounded
    number of samples of the same frame."""
    import oracle

    cores = len(os.sched_getaffinity(0))
    try:
        cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    except ValueError:
        pass
    osc = oracle.OracleScene(flat, bvh_seed=1)
    # calibrate on a strip, then size the sample to ~seconds_target
    w0, h0 = max(2, W // 8), max(2, H // 8)
    t0 = time.time()
    osc.render(cam, w0, h0, 1, sampler=oracle.OR_SAMPLER_REFERENCE, nthreads=cores)
    rate = w0 * h0 / max(time.time() - t0, 1e-3)
    spp = 1
    scale = 1.0
    budget = rate * seconds_target
    if budget < W * H:
        scale = max(0.1, (budget / (W * H)) ** 0.5)
    else:
        spp = int(min(64, max(1, budget // (W * H))))
    w1, h1 = max(2, int(W * scale)), max(2, int(H * scale))
    t0 = time.time()
    osc.render(cam, w1, h1, spp, sampler=oracle.OR_SAMPLER_REFERENCE, nthreads=cores)
    dt = time.time() - t0
    t0 = time.time()
    osc.render(cam, max(2, w1 // 4), max(2, h1 // 4), spp, sampler=oracle.OR_SAMPLER_REFERENCE, nthreads=1)
    dt1 = time.time() - t0
    return {
        "value": w1 * h1 * spp / dt / 1e6,
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{w1}x{h1} at {spp} spp of the same scene and camera, fp64 C restatement of the Scala loop "
                  f"(reference-style BVH, rejection samplers), OpenMP over rows, {dt:.1f} s",
        "value_1_thread": (max(2, w1 // 4) * max(2, h1 // 4) * spp) / dt1 / 1e6,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="dragon", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--frames-in-flight", type=int, default=1,
                    help="frames in flight, each with its own buffers and HIP stream.  2 lets the drain of one frame "
                         "overlap the start of the next, but two resident persistent launches share the CUs badly: "
                         "measured 103.2 instead of 94.9 ms per dragon frame on one GPU, so 1 is the default")
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip the CPU baseline and the host-buffer leg (profiling runs)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from raytracing_amd import _abi as A
    from raytracing_amd import lib as ht
    from raytracing_amd.distributed import FrameSlot, make_frame_step, shard_samples, timed_steps
    from raytracing_amd.scene import flatten, load_scene

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: libhiptrace has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        # build the RCCL communicator now, whatever --warmup says: its set-up takes
        # seconds and is not part of a frame
        warm = torch.zeros(1, dtype=torch.int64, device=dev)
        dist.reduce(warm, dst=0, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()

    fname, W, H, spp = WORKLOADS[args.workload]
    if args.spp > 0:
        spp = args.spp
    if args.workload == "miata":
        from raytracing_amd.meshgen import bumpy_ellipsoid
        from raytracing_amd.scene import Mesh
        cam, wl = load_scene(os.path.join(ROOT, "tests", "golden", "scenes", "dragon.json"))  # same camera, floor
        wl = [wl[0], Mesh(bumpy_ellipsoid(160, 320), wl[1].triangles[0].material)]
    else:
        cam, wl = load_scene(os.path.join(ROOT, "tests", "golden", "scenes", fname))
    flat = flatten(wl)
    scene = ht.Scene(flat, device=local_rank)
    first, count = shard_samples(spp, rank, world)

    params = ht.make_params(W, H, max(count, 1), sample_offset=first, seed=2025, batch=args.batch,
                            device=local_rank)
    launch_events = []  # (start, stop) around every render call of the timed region, on its stream

    def frame_slot():
        """Buffers and stream of one frame in flight: 32.32 fixed-point radiance sums, the
        two pictures, and a HIP stream of its own."""
        accum = torch.zeros(H * W * 3, dtype=torch.int64, device=dev)
        rgb_mean = torch.empty(H * W * 3, dtype=torch.float32, device=dev)
        packed = torch.empty(H * W, dtype=torch.int32, device=dev)
        tstream = torch.cuda.Stream(device=dev)

        def render_shard(first_, count_, timed):
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(tstream)
            # enqueue only: nothing here waits for the device, so frames overlap
            scene.render_accumulate(cam, params, accum.data_ptr(), tstream.cuda_stream, want_stats=False)
            if timed:
                e1.record(tstream)
                launch_events.append((e0, e1))

        def resolve():
            scene.resolve_device(accum.data_ptr(), W, H, spp, rgb_mean.data_ptr(), packed.data_ptr(),
                                 tstream.cuda_stream)

        return FrameSlot(accum, render_shard, resolve, lambda: torch.cuda.stream(tstream))

    # the frame loop and its timing are host logic shared with the CPU (gloo) test of the
    # multi-rank path: raytracing_amd/distributed.py
    in_flight = max(1, args.frames_in_flight)
    slots = [frame_slot() for _ in range(in_flight)]
    step = make_frame_step(spp, rank, world, slots, dist)
    elapsed = timed_steps(step, args.steps, args.warmup, world, dist, torch.cuda.synchronize,
                          lambda x: torch.tensor([x], dtype=torch.float64, device=dev))
    launches_per_call = -(-max(count, 1) // (args.batch if args.batch > 0 else 1024))
    accum = slots[0].accum
    stream = torch.cuda.current_stream().cuda_stream
    if in_flight == 1:
        # HIP events on the stream the kernel ran on, around every render call of the timed region
        kernel_s = [(e0.elapsed_time(e1) * 1e-3, launches_per_call) for e0, e1 in launch_events]
        kernel_timing = "HIP events around every launch of the timed region"
    else:
        # overlapping frames share the GPU, so a launch's own duration is not defined inside the
        # timed region: time this rank's launch alone, once more, after it
        accum.zero_()
        if count > 0:
            st = scene.render_accumulate(cam, params, accum.data_ptr(), stream, want_stats=True)
            kernel_s = [(st["seconds_kernel"], st["launches"])] * max(args.steps, 1)
        else:
            kernel_s = []
        kernel_timing = "HIP events around one launch run alone after the timed region (frames overlap inside it)"

    if rank == 0:
        total_samples = W * H * spp
        value = total_samples * args.steps / elapsed / 1e6
        # roofline of the dominant kernel (render_kernel<float>) on this rank
        ksec = sum(k for k, _ in kernel_s)
        launches = sum(n for _, n in kernel_s)
        cparams = ht.make_params(W, H, min(64, max(count, 1)), sample_offset=first, seed=2025,
                                 device=local_rank, collect_counters=1)
        accum.zero_()
        ctr = scene.render_accumulate(cam, cparams, accum.data_ptr(), stream, want_stats=True)
        flops_per_sample = algorithmic_flops(ctr) / ctr["samples"]
        rank_samples = W * H * count * args.steps
        achieved = flops_per_sample * rank_samples / max(ksec, 1e-9) / 1e12
        info = scene.info()
        # HBM bytes per launch as the PMC passes of the committed profile measured them
        # (counters cannot be read from inside this process), scaled to this run's launch
        # size; only while the profile was taken on the kernel source that runs here
        traffic, traffic_note = None, None
        try:
            import hashlib
            with open(os.path.join(ROOT, "raytracing_amd", "csrc", "kernels.hip"), "rb") as f:
                sha16 = hashlib.sha256(f.read()).hexdigest()[:16]
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                tr = json.load(f).get(args.workload)
            if not tr:
                traffic_note = "no PMC profile of this workload under profiles/"
            elif tr.get("kernels_hip_sha16") != sha16:
                traffic_note = ("profiles/traffic.json was measured on another version of kernels.hip "
                                f"({tr.get('kernels_hip_sha16')} != {sha16}): run tools/prof.sh and tools/update_traffic.py")
            elif launches:
                spp_per_launch = count * args.steps / launches
                traffic = (tr["fetch_bytes"] + tr["write_bytes"]) * spp_per_launch / tr["spp_per_launch"]
                traffic_note = ("HBM-side bytes per launch (FETCH_SIZE + WRITE_SIZE PMC passes, %s); almost all of it "
                                "is the 64-bit atomic adds that flush tile sums to the frame" % tr["source"])
        except (OSError, ValueError, KeyError) as e:
            traffic, traffic_note = None, f"traffic record unreadable: {e}"
        out = {
            "metric": "Msamples/s (camera paths per second), dragon.json 1920x1080 1024 spp"
            if args.workload == "dragon" else f"Msamples/s (camera paths per second), {args.workload}",
            "value": value,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": ("synthetic: camera and floor of scenes/miata.json with a procedural 101 760-triangle mesh in "
                     "place of the miata.stl the reference does not ship, fixed Philox seed") if args.workload == "miata"
            else "synthetic: the reference's own scene file (scenes/%s) rendered with a fixed Philox seed" % fname,
            "config": {"workload": f"{fname} {W}x{H} {spp} spp, max depth 50", "width": W, "height": H,
                       "spp": spp, "parallelism": f"spp sharded over {world} GPU(s), reduce(sum) to rank 0",
                       "frames_in_flight": in_flight,
                       "prims": info["n_prims"], "bvh_nodes": info["n_nodes"],
                       "scene_image_bytes": info["lds_bytes_fp32"],
                       "scene_store": "hbm" if info["hbm_store_fp32"] else "lds"},
            "roofline": {
                "bound": "valu_fp32",
                "achieved": achieved,
                "peak": FP32_VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP32_VALU_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_note": traffic_note,
                # the library picks the kernel per scene: render_kernel_8 (80 scalar registers, eight
                # waves per SIMD) where 32 waves fit a CU, else render_kernel_7 (96, seven waves)
                "kernel": "render_kernel_8<FEAT> or render_kernel_7<FEAT,WIDE>, whichever the library picks for the scene; "
                          "the rocprofv3 trace names it",
                "flops_per_sample": flops_per_sample,
                "kernel_ms_per_launch": ksec / max(launches, 1) * 1e3,
                "kernel_timing": kernel_timing,
                "launches_per_step": launches / max(args.steps, 1),
                "counters_per_sample": {k: (ctr[k] / ctr["samples"]) for k in
                                        ("n_rays", "n_box", "n_sphere", "n_rect", "n_tri")},
                "lane_utilisation_in_traversal": ctr["n_active"] / max(ctr["n_steps"], 1),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            # the same frame through the host-buffer entry point (ht_render): allocation of
            # the sums, render, resolve and the device-to-host copies of both pictures
            t0 = time.perf_counter()
            scene.render(cam, W, H, spp, seed=2025, batch=args.batch, device=local_rank)
            out["host_buffer_boundary"] = {"value": total_samples / (time.perf_counter() - t0) / 1e6,
                                           "unit": "Msamples/s",
                                           "note": "ht_render into host memory, PCIe copies included; never the headline value"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(flat, cam, W, H)
        print(json.dumps(out), flush=True)
    scene.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
