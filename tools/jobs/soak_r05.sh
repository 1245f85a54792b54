#!/bin/bash
# GPU job: soak of the default (faithful) preset -- divergence-guard census under the random policy and a hypothesis soak of the parity property test
mkdir -p gpurun_out
python tools/guard_census.py 600 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_guard_census.txt
LCR_HYP_EXAMPLES=400 python -m pytest tests/test_gpu_property.py -q -x 2>&1 | tail -15 > gpurun_out/r05_hyp_soak.txt
python tools/newton_const_action.py reach 4096 24 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_const_action.txt
cat gpurun_out/r05_guard_census.txt gpurun_out/r05_hyp_soak.txt; tail -4 gpurun_out/r05_const_action.txt
