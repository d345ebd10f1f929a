#!/bin/bash
# A/B of the HBM store only (run on the GPU box through gpurun): library in the tree against tools/ab_libs/*.so
set -o pipefail
OUT=gpurun_out/abm_$1
mkdir -p $OUT
for round in 1 2; do
  for lib in tree tools/ab_libs/*.so; do
    name=$(basename $lib .so)
    if [ "$lib" = tree ]; then unset HIPTRACE_LIB; else export HIPTRACE_LIB=$PWD/$lib; fi
    timeout -k 10 300 python tools/gpu_perf.py mesh:160x320 mesh:500x1000 mesh:60x120 --spp 256 --batch 256 --chunk 0 --counters 0 --wait 32 > $OUT/perf_${name}_$round.txt 2>&1 || exit 1
    echo "== $name round $round"
    grep "Msamples/s=" $OUT/perf_${name}_$round.txt | sed -E 's/ (chunk|nden|wratio|mean|lds|image|depth|kernel_s)=[^ ]*//g'
  done
done
