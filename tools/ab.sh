#!/bin/bash
# A/B helper (run on the GPU box through gpurun): kernel throughput of the benchmark scenes with the
# library in the tree and with every tools/ab_libs/*.so (other builds of it), alternating, two rounds.
#   tools/ab.sh <tag>
set -o pipefail
OUT=gpurun_out/ab_$1
mkdir -p $OUT
for round in 1 2; do
  for lib in tree tools/ab_libs/*.so; do
    name=$(basename $lib .so)
    if [ "$lib" = tree ]; then unset HIPTRACE_LIB; else export HIPTRACE_LIB=$PWD/$lib; fi
    timeout -k 10 300 python tools/gpu_perf.py dragon cornell --spp 1024 --batch 1024 --chunk 0 --counters 0 > $OUT/perf_${name}_$round.txt 2>&1 || exit 1
    timeout -k 10 300 python tools/gpu_perf.py random scene perlin checker mesh:160x320 --spp 256 --batch 256 --chunk 0 --counters 0 >> $OUT/perf_${name}_$round.txt 2>&1 || exit 2
    echo "== $name round $round"
    grep "Msamples/s=" $OUT/perf_${name}_$round.txt | sed -E 's/ (chunk|nden|wratio|wait|mean|lds|image|depth|kernel_s)=[^ ]*//g'
  done
done
