#!/bin/bash
# GPU job (round 6): the parity census, then the whole GPU suite in ONE process, as the driver runs it (-x -q)
mkdir -p gpurun_out
python tools/parity_census.py 1024 10 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_parity_census.txt
tail -3 gpurun_out/r06_parity_census.txt
(time python -m pytest tests -x -q -m gpu -p no:cacheprovider) > gpurun_out/gputests_single.log 2>&1
tail -5 gpurun_out/gputests_single.log
