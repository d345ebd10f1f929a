#!/usr/bin/env python3
"""Writes the HBM-side bytes per launch that tools/prof.sh measured (FETCH_SIZE and
WRITE_SIZE PMC passes, in KB) into profiles/traffic.json, together with the hash of the
kernel source they were measured on.  bench.py reports roofline.traffic only while that
hash matches the kernels.hip it runs.

  python tools/update_traffic.py <tag> <workload> [summary file name under profiles/]
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, workload = sys.argv[1], sys.argv[2]
source = sys.argv[3] if len(sys.argv) > 3 else f"profiles/{tag}_rocprof_summary.txt"
base = os.path.join(ROOT, "gpurun_out", "prof")


def counter(pass_no, name):
    """Average per dispatch of the production render kernel (not the counters build)."""
    tot = collections.defaultdict(float)
    files = glob.glob(f"{base}/pmc{pass_no}_{tag}/*/*_counter_collection.csv")
    # results of an earlier run with the same tag, merged into the same directory, would be summed in
    assert len(files) == 1, f"pass {pass_no}: {len(files)} result files under {base}; remove the old ones and profile again"
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "render_kernel" in k and ", true," not in k and r["Counter_Name"] == name:
                tot[r["Dispatch_Id"]] += float(r["Counter_Value"])
    # bench.py launches the production kernel once per step and once more for its statistics: the
    # same launch every time, so the figure per launch is the average over the dispatches
    assert len(tot) >= 1, f"no render launch in pass {pass_no}"
    return sum(tot.values()) / len(tot)


path = os.path.join(ROOT, "profiles", "traffic.json")
data = json.load(open(path))
data[workload] = {
    "spp_per_launch": int(open(f"{base}/spp_per_launch_{tag}.txt").read()),
    "fetch_bytes": int(counter(3, "FETCH_SIZE") * 1024),
    "write_bytes": int(counter(4, "WRITE_SIZE") * 1024),
    "kernels_hip_sha16": open(f"{base}/kernels_hip_sha16_{tag}.txt").read().strip(),
    "source": source,
}
json.dump(data, open(path, "w"), indent=2)
print(json.dumps(data[workload]))
