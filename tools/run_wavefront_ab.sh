#!/bin/bash
# persistent kernel vs wavefront variant on the on-chip and HBM scene stores, then a kernel
# trace of the wavefront variant:  tools/run_wavefront_ab.sh  (output under gpurun_out/)
cd $GRAFT_REPO_ROOT
for wf in 0 1; do
  timeout -k 10 300 python tools/gpu_perf.py dragon random cornell mesh:160x320 --spp 128 --batch 128 --counters 0 --wavefront $wf || exit 1
  timeout -k 10 300 python tools/gpu_perf.py mesh:500x1000 --spp 64 --batch 64 --counters 0 --wavefront $wf || exit 1
done
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof/wf_final -- python3 $GRAFT_REPO_ROOT/tools/gpu_perf.py mesh:160x320 dragon --spp 128 --batch 128 --counters 0 --wavefront 1 || exit 2
