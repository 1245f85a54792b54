#!/bin/bash
# GPU job: the whole GPU suite, then a hypothesis soak of the parity property test with the failing example kept
mkdir -p gpurun_out
bash tools/jobs/gpu_suite.sh
LCR_HYP_EXAMPLES=400 python -m pytest tests/test_gpu_property.py -q -x -p no:cacheprovider > gpurun_out/r05_hyp_soak_full.txt 2>&1
grep -n "Falsifying\|Failing test case" -A18 gpurun_out/r05_hyp_soak_full.txt | head -60; grep -E "^E  " gpurun_out/r05_hyp_soak_full.txt | head -12 | cut -c1-400; tail -2 gpurun_out/r05_hyp_soak_full.txt
