#!/bin/bash
# GPU job (round 6): the GPU suite under the faithful preset only, then same-box A/B of liblcr_hip_ab.so (A, the previous build) against liblcr_hip.so (B), then the per-wave phases
mkdir -p gpurun_out
(time python -m pytest tests -m gpu -q -n 4 -p no:cacheprovider -k "faithful or not (auto or single)") > gpurun_out/gputests.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/gputests.log | tail -30
bash tools/jobs/ab_times.sh reach push lift pick_place_ee 2>&1 | tee gpurun_out/ab_times.txt
python tools/newton_phases.py reach push 2>&1 | grep -v amdgpu.ids | tee gpurun_out/newton_phases.txt
