#!/bin/bash
# GPU job: a subset of the GPU suite.   bash tools/jobs/tests_subset.sh <file-with-pytest-args>   (one argument per line)
mkdir -p gpurun_out
mapfile -t ARGS < "$1"
python -m pytest "${ARGS[@]}" -q -n 4 > gpurun_out/tests_subset.log 2>&1
tail -25 gpurun_out/tests_subset.log
