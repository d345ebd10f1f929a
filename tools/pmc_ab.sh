#!/bin/bash
# Two PMC passes (vector-cache requests, fabric-side fetches) of one bench.py launch for the library
# in the tree and for every tools/ab_libs/*.so.   tools/pmc_ab.sh <tag> <workload> <spp>
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcab_$1
mkdir -p $OUT
B="python3 $R/bench.py --workload $2 --no-cpu-baseline --steps 1 --warmup 0 --spp $3"
for lib in tree $R/tools/ab_libs/*.so; do
  name=$(basename $lib .so)
  if [ "$lib" = tree ]; then unset HIPTRACE_LIB; else export HIPTRACE_LIB=$lib; fi
  timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/tcp_$name -- $B > $OUT/tcp_$name.log 2>&1 || exit 2
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/fetch_$name -- $B > $OUT/fetch_$name.log 2>&1 || exit 3
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/sq_$name -- $B > $OUT/sq_$name.log 2>&1 || exit 4
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/*_*/")):
    tot = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "render_kernel_7" in row["Kernel_Name"]:
                tot[row["Counter_Name"]] += float(row["Counter_Value"])
    print(d.rstrip("/").split("/")[-1], dict(tot))
PY
