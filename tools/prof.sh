#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + PMC passes.
#   tools/prof.sh <tag> [workload]      workload: a bench.py --workload name (default dragon)
# gpurun merges the output into the local gpurun_out/prof: remove that directory before a run
# with a tag used before, or the summaries add the old result files to the new ones.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
TAG=${1:-r01}
WL=${2:-dragon}
SPP=${3:-1024}  # samples per launch of the PMC passes: the launch the benchmark times
B="python3 $R/bench.py --workload $WL --no-cpu-baseline"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$TAG -- $B --steps 2 --warmup 1 > $OUT/trace_$TAG.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc1_$TAG -- $B --steps 1 --warmup 0 --spp $SPP > $OUT/pmc1_$TAG.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc2_$TAG -- $B --steps 1 --warmup 0 --spp $SPP > $OUT/pmc2_$TAG.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc3_$TAG -- $B --steps 1 --warmup 0 --spp $SPP > $OUT/pmc3_$TAG.log 2>&1 || exit 4
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4_$TAG -- $B --steps 1 --warmup 0 --spp $SPP > $OUT/pmc4_$TAG.log 2>&1 || exit 5
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc5_$TAG -- $B --steps 1 --warmup 0 --spp $SPP > $OUT/pmc5_$TAG.log 2>&1 || exit 6
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/pmc6_$TAG -- $B --steps 1 --warmup 0 --spp $SPP > $OUT/pmc6_$TAG.log 2>&1 || exit 7
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc7_$TAG -- $B --steps 1 --warmup 0 --spp $SPP > $OUT/pmc7_$TAG.log 2>&1 || exit 8
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU SQ_INSTS_LDS_LOAD --output-format csv -d $OUT/pmc8_$TAG -- $B --steps 1 --warmup 0 --spp $SPP > $OUT/pmc8_$TAG.log 2>&1 || exit 9
# the kernel source these counters belong to: bench.py reports roofline.traffic only while it matches
echo $SPP > $OUT/spp_per_launch_$TAG.txt
cat $R/raytracing_amd/csrc/kernels.hip $R/raytracing_amd/csrc/kernels.h $R/raytracing_amd/csrc/work_queue.h $R/raytracing_amd/csrc/scene_layout.h | sha256sum | cut -c1-16 > $OUT/kernels_hip_sha16_$TAG.txt
echo done
