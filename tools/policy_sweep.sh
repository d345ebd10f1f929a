#!/bin/bash
# Sweep of the traversal policy (run on the GPU box through gpurun): how many finished lanes a wave
# waits for before it shades (--wait) and how small a share of descending lanes keeps the leaves
# waiting (--nden).   tools/policy_sweep.sh <tag>
set -o pipefail
OUT=gpurun_out/policy_$1
mkdir -p $OUT
for nden in 2 3 4; do
  timeout -k 10 300 python tools/gpu_perf.py dragon --spp 1024 --batch 1024 --chunk 0 --counters 0 --nden $nden --wait 16 24 32 40 48 56 >> $OUT/sweep.txt 2>&1 || exit 1
  timeout -k 10 300 python tools/gpu_perf.py random mesh:160x320 --spp 256 --batch 256 --chunk 0 --counters 0 --nden $nden --wait 16 24 32 40 48 56 >> $OUT/sweep.txt 2>&1 || exit 2
done
grep "Msamples/s=" $OUT/sweep.txt | sed -E 's/ (chunk|wratio|mean|lds|image|depth|kernel_s)=[^ ]*//g'
