#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs written by tools/prof.sh into one text summary."""
import collections
import csv
import glob
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
base = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/prof"
print(f"# rocprofv3 summary, tag {tag}")
wl = sys.argv[3] if len(sys.argv) > 3 else "dragon"
print(f"\n## kernel trace (--kernel-trace --stats): bench.py --workload {wl} --steps 2 --warmup 1 --no-cpu-baseline")
for f in glob.glob(f"{base}/trace_{tag}/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:90]:90s} calls={r['Calls']:>4s} total_ms={float(r['TotalDurationNs'])/1e6:10.3f} "
              f"avg_ms={float(r['AverageNs'])/1e6:9.4f} pct={float(r['Percentage']):6.2f}")
print(f"\n## PMC passes (one rocprofv3 --pmc run each): bench.py --workload {wl} --steps 1 --warmup 0 --spp <samples per launch, default 1024> --no-cpu-baseline")
print("## values are averages per dispatch of the kernel in that run (bench.py launches the production kernel once per")
print("## step and once more for its statistics; the counters build, render_kernel<float, true, ..>, runs 64 spp once)")
for n in (1, 2, 3, 4, 5, 6, 7, 8):
    for f in glob.glob(f"{base}/pmc{n}_{tag}/*/*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        meta = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "render_kernel" not in k and "resolve_kernel" not in k:
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
            meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"],
                       r["Workgroup_Size"], r["Grid_Size"])
        for k, v in agg.items():
            m = meta[k]
            print(f"pass {n}: {k[:70]} dispatches={len(disp[k])} vgpr={m[0]} agpr={m[1]} sgpr={m[2]} lds={m[3]} "
                  f"scratch={m[4]} wg={m[5]} grid={m[6]}")
            for c, x in sorted(v.items()):
                print(f"    {c:28s} {x / len(disp[k]):.6g}")
