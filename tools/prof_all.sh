#!/bin/bash
# tools/prof.sh for several benchmarked workloads (run on the GPU box through gpurun).
#   tools/prof_all.sh <tag> <workload:spp per launch> ...
set -o pipefail
T=$1; shift
R=$GRAFT_REPO_ROOT
for wl in "$@"; do
  w=${wl%%:*}; spp=${wl##*:}
  echo "== profiling $w ($spp spp per launch)"
  bash $R/tools/prof.sh ${T}_$w $w $spp || { echo "prof.sh failed for $w: $?"; exit 1; }
done
echo profiles done
