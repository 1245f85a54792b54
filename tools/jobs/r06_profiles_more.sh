#!/bin/bash
# GPU job (round 6): bench line + rocprofv3 stats + PMC under the shipped default for the workloads BASELINE's configs do not name: PushCubeLoop-v0, LiftCube-v0, StackTwoCubes-v0 state-only at 65 536 envs
mkdir -p gpurun_out
bash tools/profile_gpu.sh r06_loop --workload PushCubeLoop-v0 > /dev/null 2>&1
bash tools/profile_gpu.sh r06_lift --workload LiftCube-v0 > /dev/null 2>&1
bash tools/profile_gpu.sh r06_stack --workload StackTwoCubes-v0 > /dev/null 2>&1
for t in r06_loop r06_lift r06_stack; do head -c 300 gpurun_out/prof_$t/bench_line.json; echo; done
