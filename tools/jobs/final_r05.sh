#!/bin/bash
# GPU job: the round-5 evidence set (profiles/r05_*): bench line + rocprofv3 stats + PMC of the default, step times of every workload under both presets, per-wave cycles, tail census; config 5 (frames) profile and the frame kernel's own time / work / counters
mkdir -p gpurun_out
bash tools/profile_gpu.sh r05 > /dev/null 2>&1
python tools/summarize_profile.py r05 r05 > gpurun_out/r05_summary.log 2>&1
(python tools/quick_times.py --steps 200; python tools/quick_times.py --steps 200 --preset fast) 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_quick_times.txt
python tools/rail_census.py 400 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_rail_census.txt
python tools/newton_dev_check.py reach,push,lift,pick_place,stack,push_loop 4096 6 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_newton_parity.txt
bash tools/profile_gpu.sh r05_c5 --config 5 --preset fast > /dev/null 2>&1
bash tools/jobs/render_check.sh > gpurun_out/r05_render.txt 2>&1
bash tools/jobs/render_pmc.sh >> gpurun_out/r05_render.txt 2>&1
