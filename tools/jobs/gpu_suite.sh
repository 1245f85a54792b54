#!/bin/bash
# GPU job: the whole GPU suite (no -x: every failure is listed)
mkdir -p gpurun_out
(time python -m pytest tests -m gpu -q -n 4 -p no:cacheprovider) > gpurun_out/gputests.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/gputests.log | tail -40
