#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + separate PMC passes of the bench command.
# Usage: tools/profile_gpu.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 50 --warmup 5 --no-cpu-baseline --calibrate 20 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o p -- $BENCH > $OUT/pmc_$C.log 2>&1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_SQ -o p -- $BENCH > $OUT/pmc_SQ.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_WAVES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_SQ2 -o p -- $BENCH > $OUT/pmc_SQ2.log 2>&1
find $OUT -name "*.csv" | head -40
# config-5 shape: Stack + ray-cast frames (HBM-write-bound), kernel trace only (tools/profile_render.sh adds its PMC passes)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_img -o t -- python $REPO/bench.py --workload StackTwoCubes-v0 --obs both --envs-per-gpu 32768 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/trace_img.log 2>&1
# the other BASELINE.json workloads at 65 536 envs (kernel trace only)
for WL in PushCube-v0 LiftCube-v0 PickPlaceCube-v0 StackTwoCubes-v0 PushCubeLoop-v0; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$WL -o t -- python $REPO/bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline > $OUT/trace_$WL.log 2>&1
done
