#!/bin/bash
# chunk-size scan (work-unit size) on the benchmark frames: tools/run_chunks.sh
cd $GRAFT_REPO_ROOT
for c in 8 16 32; do
  timeout -k 10 200 python tools/gpu_perf.py dragon cornell --spp 1024 --batch 1024 --counters 0 --chunk $c --ctail 4 || exit 1
done
