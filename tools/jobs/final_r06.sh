#!/bin/bash
# GPU job: the round-6 evidence set (profiles/r06_*): bench line + rocprofv3 stats + PMC of the default on BASELINE configs 2 (headline), 3, 4, 5 -- all under the shipped
# default preset --, step times of every workload under both presets, per-wave phases of the Newton kernels
mkdir -p gpurun_out
bash tools/profile_gpu.sh r06 > /dev/null 2>&1
bash tools/profile_gpu.sh r06_c3 --config 3 > /dev/null 2>&1
bash tools/profile_gpu.sh r06_c4 --config 4 > /dev/null 2>&1
bash tools/profile_gpu.sh r06_c5 --config 5 > /dev/null 2>&1
(python tools/quick_times.py --steps 200; python tools/quick_times.py --steps 200 --preset fast) 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_quick_times.txt
python tools/newton_phases.py reach push lift pick_place_ee stack 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_newton_phases.txt
python tools/newton_phases.py stack --n 32768 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_newton_phases.txt
for t in r06 r06_c3 r06_c4 r06_c5; do head -c 400 gpurun_out/prof_$t/bench_line.json; echo; done
python tools/newton_dev_check.py push,lift,stack 2048 16 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_newton_dev_check.txt
cat gpurun_out/r06_newton_dev_check.txt
