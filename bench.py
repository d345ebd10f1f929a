#!/usr/bin/env python3
"""Headline benchmark: Msamples/s (camera paths per second) of the trace /
intersect / shade loop on scenes/dragon.json at 1920x1080, 1024 spp.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With --gpus N > 1 and no WORLD_SIZE in the environment the script starts the N ranks itself
(torch.distributed.run as a child process) and relays their output, so both command lines
above measure N GPUs; the JSON line carries "ranks_seen", the size of the process group that
really ran, and the run fails when it differs from --gpus.

One "step" = one whole frame, fixed while N grows (strong scaling).  Two ways to cut it:
  --shard tiles (default for N > 1): rank g renders every sample of the 8x8 tiles g, g + N, ...
      into its own buffer of radiance sums (int64, 32.32 fixed point), resolves its own tiles
      (mean, tone-map) and rank 0 gathers the packed tile buffers (RCCL over xGMI, 4 bytes per
      pixel in all) and puts the picture together;
  --shard samples: rank g renders its share of the samples-per-pixel range for all pixels, the
      sum buffers are added onto rank 0 with one reduce (24 bytes per pixel) and rank 0 resolves.
Both give the one-GPU frame bit for bit.  Rank 0 prints ONE JSON line.

Timed region: scene already resident on the device; barrier +
torch.cuda.synchronize() on both sides; max over ranks.
"""
import argparse
import json
import os
import socket
import subprocess
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
# 256 CU x 128 lanes x 2 flop x 2.4 GHz; fp64 vector rate is half of it
VALU_PEAK_TFLOPS = {"fp32": 157.3, "fp64": 78.6}


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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="dragon", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--shard", default="tiles", choices=("tiles", "samples"),
                    help="how N > 1 ranks cut the frame: interleaved 8x8 tiles (no sums travel, rank 0 gathers the "
                         "packed pictures) or ranges of the samples per pixel (one reduce of the int64 sums)")
    ap.add_argument("--precision", default="fp32", choices=("fp32", "fp64"),
                    help="arithmetic of the device path: fp32 (production) or fp64 (the reference's own arithmetic)")
    ap.add_argument("--frames-in-flight", type=int, default=1,
                    help="frames in flight, each with its own buffers and HIP stream.  2 lets the drain of one frame "
                         "overlap the start of the next, but two resident persistent launches share the CUs badly: "
                         "measured 103.2 instead of 94.9 ms per dragon frame on one GPU, so 1 is the default")
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE",
                    help="a library option (ht_set_option) for this run, e.g. --set xcd_bands=0 (profiling A/B)")
    ap.add_argument("--rehearse-collectives", action="store_true",
                    help="with one GPU: build a one-rank RCCL process group and run the frame's gather (--shard tiles) "
                         "or reduce (--shard samples), the barrier and the all-reduce of the timing through it, as an "
                         "N-rank run would; the JSON line says so in 'rehearsal'")
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip the CPU baseline and the host-buffer leg (profiling runs)")
    ap.add_argument("--selftest-standin", action="store_true",
                    help="launcher self-test, no GPU: the ranks run this script's frame loop over gloo on CPU tensors "
                         "with a stand-in renderer that has a closed-form frame; the JSON line is not a measurement")
    return ap.parse_args(argv)


def launch_ranks(args, argv):
    """Start --gpus ranks of this script as a child process (torch.distributed.run, one rank per
    GPU), relay what they print, and return the child's exit code.  Nothing in this process has
    touched the GPU: the ranks own it."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC between the ranks' processes
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def standin_rank(args, world, rank):
    """The launcher self-test: this script's frame loop (raytracing_amd.distributed) over gloo on
    CPU tensors.  The stand-in renderer adds a known integer per (pixel, global sample index), so
    the frame has a closed form and any lost, doubled or misplaced sample or tile shows."""
    import torch
    import torch.distributed as dist

    from raytracing_amd.distributed import FrameSlot, make_frame_step, shard_samples, shard_tiles, timed_steps

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, spp = 40, 24, 7
    tiles_x, tiles_y = (W + 7) // 8, (H + 7) // 8
    n_tiles = tiles_x * tiles_y
    mode = args.shard if world > 1 else "samples"
    t_first, t_stride = shard_tiles(rank, world) if mode == "tiles" else (0, 1)
    shard_px = -(-n_tiles // world) * 64
    pix = torch.arange(H * W, dtype=torch.int64)
    tile_of = (pix // W // 8) * tiles_x + (pix % W) // 8
    mine = (tile_of % t_stride) == t_first

    def value(s_):  # what sample s_ adds to every pixel's sum
        return pix * 1000 + s_ * s_ + 1

    accum = torch.zeros(H * W, dtype=torch.int64)
    picture = torch.zeros(H * W, dtype=torch.int64)
    tiles = torch.zeros(shard_px, dtype=torch.int64)
    tiles_all = torch.zeros(shard_px * world, dtype=torch.int64) if rank == 0 else None
    # pixel p lies in tile buffer slot (tile // stride) * 64 + (row in tile) * 8 + column of its owner
    slot_of = (tile_of // t_stride) * 64 + (pix // W % 8) * 8 + pix % W % 8

    def render_shard(first, count, timed):
        for s_ in range(first, first + count):
            accum.add_(torch.where(mine, value(s_), torch.zeros_like(pix)))

    def resolve():
        picture.copy_(accum)

    def resolve_tiles():
        tiles.zero_()
        tiles[slot_of[mine]] = accum[mine]

    def assemble():
        owner = tile_of % world
        src = owner * shard_px + (tile_of // world) * 64 + (pix // W % 8) * 8 + pix % W % 8
        picture.copy_(tiles_all[src])

    slot = FrameSlot(accum, render_shard, resolve, tiles=tiles, tiles_all=tiles_all, resolve_tiles=resolve_tiles,
                     assemble=assemble)
    step = make_frame_step(spp, rank, world, [slot], dist, mode=mode)
    elapsed = timed_steps(step, args.steps, args.warmup, world, dist, lambda: None,
                          lambda x: torch.tensor([x], dtype=torch.float64))
    if rank == 0:
        whole = sum(value(s_) for s_ in range(spp))
        print(json.dumps({
            "metric": "launcher self-test (stand-in renderer on CPU tensors over gloo): not a measurement",
            "selftest": True, "value": 0.0, "unit": "Msamples/s", "n_gpus": world,
            "ranks_seen": dist.get_world_size() if world > 1 else 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "shard": mode,
            "frame_equals_closed_form": bool(torch.equal(picture, whole)),
            "data": "stand-in", "config": {"workload": f"closed-form frame {W}x{H} {spp} spp"},
        }), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (before anything touches the GPU)
        raise SystemExit(launch_ranks(args, argv))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the process group would not be the one asked for")
    if args.selftest_standin:
        return standin_rank(args, world, rank)

    import torch
    import torch.distributed as dist

    from raytracing_amd import _abi as A
    from raytracing_amd import lib as ht
    from raytracing_amd.distributed import FrameSlot, make_frame_step, shard_samples, shard_tiles, timed_steps
    from raytracing_amd.scene import flatten, load_scene

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: libhiptrace has no CPU path")
    for kv in args.set:
        name, _, val = kv.partition("=")
        ht.set_option(name, int(val))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks_seen = 1
    rehearse = args.rehearse_collectives and world == 1
    if rehearse:  # a process group of one rank, set up like a launcher would
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
    if world > 1 or rehearse:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        ranks_seen = dist.get_world_size()
        if ranks_seen != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {ranks_seen} ranks")
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
    precision = A.HT_PRECISION_FP64 if args.precision == "fp64" else A.HT_PRECISION_FP32
    mode = args.shard if (world > 1 or rehearse) else "samples"
    if mode == "tiles":
        first, count = 0, spp                      # every sample of this rank's own tiles
        t_first, t_stride = shard_tiles(rank, world)
    else:
        first, count = shard_samples(spp, rank, world)
        t_first, t_stride = 0, 1
    shard_px = ht.shard_tile_pixels(W, H, world)

    params = ht.make_params(W, H, max(count, 1), sample_offset=first, seed=2025, batch=args.batch,
                            device=local_rank, precision=precision, tile_first=t_first, tile_stride=t_stride)
    launch_events = []  # (start, stop) around every render call of the timed region, on its stream

    def frame_slot():
        """Buffers and stream of one frame in flight: 32.32 fixed-point radiance sums, the
        pictures, and a HIP stream of its own."""
        accum = torch.zeros(H * W * 3, dtype=torch.int64, device=dev)
        rgb_mean = torch.empty(H * W * 3, dtype=torch.float32, device=dev)
        packed = torch.empty(H * W, dtype=torch.int32, device=dev)
        tstream = torch.cuda.Stream(device=dev)
        tiles = mean_tiles = tiles_all = None
        if mode == "tiles":
            tiles = torch.zeros(shard_px, dtype=torch.int32, device=dev)
            mean_tiles = torch.zeros(shard_px * 3, dtype=torch.float32, device=dev)
            if rank == 0:
                tiles_all = torch.zeros(shard_px * world, dtype=torch.int32, device=dev)

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

        def resolve_tiles():  # this rank's own tiles: means stay here, the packed codes are gathered
            scene.resolve_tiles_device(accum.data_ptr(), W, H, spp, t_first, t_stride, mean_tiles.data_ptr(),
                                       tiles.data_ptr(), tstream.cuda_stream)

        def assemble():
            scene.assemble_tiles_device(W, H, world, 0, tiles_all.data_ptr(), 0, packed.data_ptr(), tstream.cuda_stream)

        slot = FrameSlot(accum, render_shard, resolve, lambda: torch.cuda.stream(tstream), tiles=tiles,
                         tiles_all=tiles_all, resolve_tiles=resolve_tiles, assemble=assemble)
        slot.packed = packed  # the picture the frame loop leaves on rank 0
        return slot

    # the frame loop and its timing are host logic shared with the CPU (gloo) test of the
    # multi-rank path: raytracing_amd/distributed.py
    in_flight = max(1, args.frames_in_flight)
    slots = [frame_slot() for _ in range(in_flight)]
    step = make_frame_step(spp, rank, world, slots, dist, mode=mode, collectives_at_one=rehearse)
    elapsed = timed_steps(step, args.steps, args.warmup, world, dist, torch.cuda.synchronize,
                          lambda x: torch.tensor([x], dtype=torch.float64, device=dev), collectives_at_one=rehearse)
    accum = slots[0].accum
    stream = torch.cuda.current_stream().cuda_stream
    rehearsal_ok = None
    if rehearse:
        # the picture that came through the collectives against one resolved straight from the sums
        torch.cuda.synchronize()
        direct = torch.empty(H * W, dtype=torch.int32, device=dev)
        scene.resolve_device(accum.data_ptr(), W, H, spp, 0, direct.data_ptr(), stream)
        torch.cuda.synchronize()
        rehearsal_ok = bool(torch.equal(direct, slots[0].packed)) and bool(direct.any())
    # One more call of the same launch, alone and with statistics: the library reports the kernel
    # it chose, the samples per launch it planned and the number of launches it made.
    accum.zero_()
    probe = scene.render_accumulate(cam, params, accum.data_ptr(), stream, want_stats=True) if count > 0 else None
    launches_per_call = probe["launches"] if probe else 0
    if in_flight == 1:
        # HIP events on the stream the kernel ran on, around every render call of the timed region
        kernel_s = [(e0.elapsed_time(e1) * 1e-3, launches_per_call) for e0, e1 in launch_events]
        kernel_timing = "HIP events around every launch of the timed region"
    else:
        # overlapping frames share the GPU, so a launch's own duration is not defined inside the
        # timed region: this rank's launch timed alone, after it
        kernel_s = [(probe["seconds_kernel"], probe["launches"])] * max(args.steps, 1) if probe else []
        kernel_timing = "HIP events around one launch run alone after the timed region (frames overlap inside it)"

    if rank == 0:
        total_samples = W * H * spp
        value = total_samples * args.steps / elapsed / 1e6
        # roofline of the dominant kernel on this rank
        ksec = sum(k for k, _ in kernel_s)
        launches = sum(n for _, n in kernel_s)
        cparams = ht.make_params(W, H, min(64, max(count, 1)), sample_offset=first, seed=2025,
                                 device=local_rank, collect_counters=1, precision=precision,
                                 tile_first=t_first, tile_stride=t_stride)
        accum.zero_()
        ctr = scene.render_accumulate(cam, cparams, accum.data_ptr(), stream, want_stats=True)
        # ht_stats.samples counts the paths the call traced: this rank's pixels (all of them, or those
        # of its own tiles) x the samples per pixel of the call
        flops_per_sample = algorithmic_flops(ctr) / max(ctr["samples"], 1)
        rank_pixels = ctr["samples"] // max(ctr["samples_per_pixel"], 1)
        rank_samples = rank_pixels * count * args.steps
        achieved = flops_per_sample * rank_samples / max(ksec, 1e-9) / 1e12
        peak = VALU_PEAK_TFLOPS[args.precision]
        info = scene.info()
        # HBM bytes per launch as the PMC passes of the committed profile measured them
        # (counters cannot be read from inside this process), scaled to this run's launch
        # size; only while the profile was taken on the kernel source that runs here
        traffic, traffic_note = None, None
        try:
            import hashlib
            h = hashlib.sha256()  # the device code and what it is launched with (tools/prof.sh hashes the same files)
            for name in ("kernels.hip", "kernels.h", "work_queue.h", "scene_layout.h"):
                with open(os.path.join(ROOT, "raytracing_amd", "csrc", name), "rb") as f:
                    h.update(f.read())
            sha16 = h.hexdigest()[:16]
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                tr = json.load(f).get(args.workload)
            if args.precision != "fp32":
                traffic_note = "the PMC profiles under profiles/ are of the fp32 kernels"
            elif not tr:
                traffic_note = "no PMC profile of this workload under profiles/"
            elif tr.get("kernels_hip_sha16") != sha16:
                traffic_note = ("profiles/traffic.json was measured on another version of the kernel sources "
                                f"({tr.get('kernels_hip_sha16')} != {sha16}): run tools/prof.sh and tools/update_traffic.py")
            elif launches:
                spp_per_launch = count * args.steps / launches
                traffic = ((tr["fetch_bytes"] + tr["write_bytes"]) * spp_per_launch / tr["spp_per_launch"]
                           * rank_pixels / (W * H))
                traffic_note = ("HBM-side bytes per launch (FETCH_SIZE + WRITE_SIZE PMC passes, %s); on the on-chip "
                                "scenes almost all of it is the 64-bit atomic adds that flush tile sums to the frame"
                                % tr["source"])
        except (OSError, ValueError, KeyError) as e:
            traffic, traffic_note = None, f"traffic record unreadable: {e}"
        parallelism = (f"8x8 tiles dealt round-robin to {world} GPU(s), every sample of a tile on its owner, "
                       "gather of the packed tile buffers to rank 0" if mode == "tiles" else
                       f"spp sharded over {world} GPU(s), reduce(sum) of the int64 sums to rank 0")
        out = {
            "metric": "Msamples/s (camera paths per second), dragon.json 1920x1080 1024 spp"
            if args.workload == "dragon" else f"Msamples/s (camera paths per second), {args.workload}",
            "value": value,
            "unit": "Msamples/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "rehearsal": ("one-rank RCCL group: every frame ran its %s, the barriers and the all-reduce of the timing "
                          "through the collective calls of an N-rank run" % ("gather of packed tiles" if mode == "tiles"
                                                                          else "reduce of the int64 sums")) if rehearse else None,
            "rehearsal_picture_equals_direct_resolve": rehearsal_ok,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64" if args.precision == "fp64" else "f32",
            "data": ("synthetic: camera and floor of scenes/miata.json with a procedural 101 760-triangle mesh in "
                     "place of the miata.stl the reference does not ship, fixed Philox seed") if args.workload == "miata"
            else "synthetic: the reference's own scene file (scenes/%s) rendered with a fixed Philox seed" % fname,
            "config": {"workload": f"{fname} {W}x{H} {spp} spp, max depth 50", "width": W, "height": H,
                       "spp": spp, "parallelism": parallelism, "shard": mode,
                       "frames_in_flight": in_flight,
                       "prims": info["n_prims"], "bvh_nodes": info["n_nodes"],
                       "scene_image_bytes": info["lds_bytes_fp64" if args.precision == "fp64" else "lds_bytes_fp32"],
                       "scene_store": "hbm" if info["hbm_store_fp64" if args.precision == "fp64" else "hbm_store_fp32"]
                       else "lds"},
            "roofline": {
                "bound": "valu_" + args.precision,
                "achieved": achieved,
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": achieved / peak,
                "traffic": traffic,
                "traffic_note": traffic_note,
                # what the library reports it launched (ht_stats.kernel): the name a rocprofv3 trace shows
                "kernel": probe["kernel"] if probe else None,
                "kernel_sgpr_budget": probe["kernel_sgprs"] if probe else None,
                "launch_shape": ({"block_threads": probe["block_threads"], "blocks_per_cu": probe["blocks_per_cu"],
                                  "lds_bytes": probe["lds_bytes"]} if probe else None),
                "flops_per_sample": flops_per_sample,
                "kernel_ms_per_launch": ksec / max(launches, 1) * 1e3,
                "kernel_timing": kernel_timing,
                "launches_per_step": launches / max(args.steps, 1),
                "samples_per_launch_planned": probe["batch"] if probe else None,
                "counters_per_sample": {k: (ctr[k] / max(ctr["samples"], 1)) for k in
                                        ("n_rays", "n_box", "n_sphere", "n_rect", "n_tri")},
                "lane_utilisation_in_traversal": ctr["n_active"] / max(ctr["n_steps"], 1),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            # The same frame through the host-buffer entry point (ht_render), which is what a
            # renderFrame replacement calls: render, resolve and the device-to-host copies.  The
            # first call grows the handle's cached device buffers; later calls allocate nothing.
            import numpy as np
            import ctypes as C

            def host_call(want_linear, calls=1, pinned=False):
                p = ht.make_params(W, H, spp, seed=2025, batch=args.batch, device=local_rank, precision=precision)
                # written once before the call so that the copy does not pay the caller's first-touch page faults
                # (a renderFrame replacement fills the same BufferedImage array frame after frame)
                rgb = np.full((H, W, 3), 0.5, dtype=np.float32) if want_linear else None
                pk = np.full((H, W), 1, dtype=np.uint32)
                if pinned:  # page-locked caller memory: the copy runs at the link's rate, without staging
                    keep = torch.ones((H, W), dtype=torch.int32).pin_memory()
                    pk = keep.numpy().view(np.uint32)
                cc = cam.to_c()
                best = None
                for _ in range(calls):
                    t0 = time.perf_counter()
                    rc = ht.load().ht_render(scene.handle, C.byref(cc), C.byref(p),
                                             rgb.ctypes.data_as(A._pf32) if want_linear else None,
                                             pk.ctypes.data_as(A._pu32), None)
                    dt = time.perf_counter() - t0
                    if rc != 0:
                        raise RuntimeError(ht.load().ht_last_error().decode())
                    best = dt if best is None else min(best, dt)
                return total_samples / best / 1e6

            first_call = host_call(True)
            out["host_buffer_boundary"] = {
                "value": host_call(False, 3),
                "value_with_linear_means": host_call(True, 3),
                "value_into_pinned_memory": host_call(False, 3, pinned=True),
                "value_first_call": first_call,
                "unit": "Msamples/s",
                "note": "ht_render into host memory, PCIe copies included, best of 3 calls; never the headline value.  value: the packed "
                        "0x00RRGGBB picture only, which is all renderFrame's BufferedImage receives (main.scala:93); "
                        "value_with_linear_means: also the 12-byte-per-pixel float means; value_into_pinned_memory: the packed picture into "
                        "page-locked memory of the caller's; value_first_call: the call "
                        "that allocates the handle's cached device buffers"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(flat, cam, W, H)
        print(json.dumps(out), flush=True)
    scene.close()
    if world > 1 or rehearse:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
