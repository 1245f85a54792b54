#!/bin/bash
# GPU job: hypothesis soak of the parity property test (LCR_HYP_EXAMPLES examples, default 1500); the failing example, if any, is kept
mkdir -p gpurun_out
LCR_HYP_EXAMPLES=${1:-1500} python -m pytest tests/test_gpu_property.py -q -x -p no:cacheprovider > gpurun_out/r05_hyp_soak_full.txt 2>&1
grep -n "Failing test case" -A18 gpurun_out/r05_hyp_soak_full.txt | head -40; grep -E "^E  " gpurun_out/r05_hyp_soak_full.txt | head -6 | cut -c1-400; tail -2 gpurun_out/r05_hyp_soak_full.txt
