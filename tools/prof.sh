#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + PMC passes.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
TAG=${1:-r01}
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$TAG -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/trace_$TAG.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc1_$TAG -- python3 $R/bench.py --steps 1 --warmup 0 --spp 256 --no-cpu-baseline > $OUT/pmc1_$TAG.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc2_$TAG -- python3 $R/bench.py --steps 1 --warmup 0 --spp 256 --no-cpu-baseline > $OUT/pmc2_$TAG.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc3_$TAG -- python3 $R/bench.py --steps 1 --warmup 0 --spp 256 --no-cpu-baseline > $OUT/pmc3_$TAG.log 2>&1 || exit 4
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4_$TAG -- python3 $R/bench.py --steps 1 --warmup 0 --spp 256 --no-cpu-baseline > $OUT/pmc4_$TAG.log 2>&1 || exit 5
echo done
