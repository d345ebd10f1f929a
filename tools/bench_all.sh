#!/bin/bash
# A bench.py record of every workload plus the fp64 dragon (run on the GPU box through gpurun).
#   tools/bench_all.sh <tag>
set -o pipefail
T=$1
mkdir -p gpurun_out/bench_$T
for w in dragon random cornell perlin checker scene miata; do
  echo "== bench $w"
  timeout -k 10 300 python bench.py --workload $w > gpurun_out/bench_$T/$w.json 2> gpurun_out/bench_$T/$w.err || { echo "bench failed for $w"; tail -5 gpurun_out/bench_$T/$w.err; exit 2; }
done
timeout -k 10 300 python bench.py --workload dragon --precision fp64 --spp 128 > gpurun_out/bench_$T/dragon_fp64.json 2> gpurun_out/bench_$T/dragon_fp64.err || { tail -5 gpurun_out/bench_$T/dragon_fp64.err; exit 3; }
echo benches done
