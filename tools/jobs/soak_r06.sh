#!/bin/bash
# GPU job (round 6): soak of the default preset with the cooperative solves -- hypothesis soak of the parity property test, divergence-guard census under the random policy
mkdir -p gpurun_out
LCR_HYP_EXAMPLES=${1:-800} python -m pytest tests/test_gpu_property.py -q -x -p no:cacheprovider > gpurun_out/r06_hyp_soak_full.txt 2>&1
(grep -n "Failing test case" -A18 gpurun_out/r06_hyp_soak_full.txt | head -40; grep -E "^E  " gpurun_out/r06_hyp_soak_full.txt | head -6 | cut -c1-400; tail -2 gpurun_out/r06_hyp_soak_full.txt) > gpurun_out/r06_hyp_soak.txt
python tools/guard_census.py 400 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_guard_census.txt
cat gpurun_out/r06_hyp_soak.txt gpurun_out/r06_guard_census.txt
python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "test_link_proxy_contacts and faithful" 2>&1 | grep -E "^FAILED|^E  .*assert|passed|failed" | cut -c1-300 > gpurun_out/r06_tight_link.txt
cat gpurun_out/r06_tight_link.txt
python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "cooperative_and_simt" 2>&1 | grep -E "^(FAILED|ERROR)|^E  .*assert|passed|failed" | cut -c1-300 | tee gpurun_out/r06_layouts_agree.txt
