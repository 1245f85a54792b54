#!/bin/bash
# GPU job (round 6): the whole GPU suite, the CPU oracle's thread scaling on this host, A/B step times
mkdir -p gpurun_out
(time python -m pytest tests -m gpu -q -n 4 -p no:cacheprovider) > gpurun_out/gputests.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/gputests.log | tail -30
python tools/cpu_scaling.py reach 2>&1 | grep -v amdgpu.ids | tee gpurun_out/cpu_scaling.txt
bash tools/jobs/ab_times.sh reach push 2>&1 | tee gpurun_out/ab_times.txt
