#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + separate PMC passes of one bench command.
# Usage: tools/profile_gpu.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/   (then tools/summarize_profile.py <tag> <round> ...)
#   e.g. tools/profile_gpu.sh r02                      (bench default = BASELINE config 2)
#        tools/profile_gpu.sh r02_c3 --config 3
# --pmc passes are separate runs with --kernel-trace only (never combined with sys/runtime tracing).
set -u
TAG=${1:-r02}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the bench line of the plain command (no profiler attached), default step counts
python $REPO/bench.py $* > $OUT/bench_line.json 2> $OUT/bench_line.err
BENCH="python $REPO/bench.py --steps 50 --warmup 5 --repeats 1 --no-cpu-baseline --no-fast-side --calibrate 20 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o p -- $BENCH > $OUT/pmc_$C.log 2>&1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_SQ -o p -- $BENCH > $OUT/pmc_SQ.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_WAVES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_SQ2 -o p -- $BENCH > $OUT/pmc_SQ2.log 2>&1
find $OUT -name "*.csv" | head -20
