#!/bin/bash
# Evidence for the scalar-register limit on resident waves: the Cornell box (no LDS image) at
# forced wave limits.  With "max_waves_per_cu" = 32 the shape search ties between the two
# kernels at 256 x 8 and keeps render_kernel_7 (96 scalar registers), whose eighth wave per
# SIMD must not be resident if the rule holds: same rate as 256 x 7.
cd $GRAFT_REPO_ROOT
for mw in 24 28 32 0; do
  timeout -k 10 100 python tools/gpu_perf.py cornell mesh:4x6 --maxwaves $mw --spp 1024 --batch 1024 --chunk 0 --counters 0 2>&1 | grep Msamples | sed -E "s/chunk=.*kernel_s/max_waves_per_cu=$mw kernel_s/; s/mean=\S+//"
done
cd /tmp && export TMPDIR=/tmp
for mw in 32 0; do
timeout -s KILL 100 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof/sgpr_$mw -- python3 $GRAFT_REPO_ROOT/tools/gpu_perf.py cornell --maxwaves $mw --spp 1024 --batch 1024 --chunk 0 --counters 0 > /dev/null 2>&1
grep -h "render_kernel" $GRAFT_REPO_ROOT/gpurun_out/prof/sgpr_$mw/*/*_kernel_stats.csv | cut -d, -f1-4 | sed "s/^/max_waves_per_cu=$mw: /"
done
