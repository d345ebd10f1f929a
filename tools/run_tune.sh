#!/bin/bash
# development: scheduling knobs on the dragon and the Cornell box at the current launch shapes
for nden in 2 3; do
  timeout -k 10 200 python tools/gpu_perf.py dragon cornell --nden $nden --wait 32 36 40 44 48 56 --spp 1024 --batch 1024 --chunk 0 --counters 0 2>&1 | grep Msamples | sed -E "s/chunk=.*nden/nden/; s/kernel_s=\S+//; s/mean=.*//"
done
for ck in 8 16 32; do
  timeout -k 10 200 python tools/gpu_perf.py dragon cornell --chunk $ck --spp 1024 --batch 1024 --counters 0 2>&1 | grep Msamples | sed -E "s/nden=.*kernel_s=\S+//; s/mean=.*//"
done
