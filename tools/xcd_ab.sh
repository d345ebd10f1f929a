#!/bin/bash
# HBM store with and without the banded work queue (option "xcd_bands"), alternating, two rounds, then the
# fabric-side fetch counters of one benchmark launch for both (run on the GPU box through gpurun).
#   tools/xcd_ab.sh <tag>
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/xcd_$1
mkdir -p $OUT
for round in 1 2; do
  for x in 1 0; do
    timeout -k 10 300 python tools/gpu_perf.py mesh:160x320 mesh:500x1000 mesh:60x120 --spp 256 --batch 256 --chunk 0 --counters 0 --wait 32 --xcd $x > $OUT/perf_xcd${x}_$round.txt 2>&1 || exit 1
    echo "== xcd_bands $x round $round"
    grep "Msamples/s=" $OUT/perf_xcd${x}_$round.txt | sed -E 's/ (chunk|nden|wratio|mean|lds|image|depth|kernel_s)=[^ ]*//g'
  done
done
cd /tmp && export TMPDIR=/tmp
for x in 1 0; do
  B="python3 $R/bench.py --workload miata --no-cpu-baseline --steps 1 --warmup 0 --set xcd_bands=$x"
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/fetch_$x -- $B > $OUT/fetch_$x.log 2>&1 || exit 3
  timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/tcc_$x -- $B > $OUT/tcc_$x.log 2>&1 || exit 4
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/sq_$x -- $B > $OUT/sq_$x.log 2>&1 || exit 5
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/*_[01]/")):
    tot = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "render_kernel_7" in row["Kernel_Name"]:
                tot[row["Counter_Name"]] += float(row["Counter_Value"])
    print(d.rstrip("/").split("/")[-1], dict(tot))
PY
