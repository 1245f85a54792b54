#!/bin/bash
# frame kernel: tests of the image paths, its time on the config-5 shape (StackTwoCubes, 32 768 envs) and on ReachCube, and its ray-cast work per env
bash tools/jobs/tests_subset.sh tools/jobs/subset_f.txt | grep -E "FAILED|passed|failed|Error"
python tools/render_time.py 2>&1 | grep -v amdgpu.ids | tail -1
python tools/render_time.py 32768 reach 2>&1 | grep -v amdgpu.ids | tail -1
LCR_RENDER_COUNT=1 python tools/render_work.py stack 2>&1 | tail -1
LCR_RENDER_COUNT=1 python tools/render_work.py reach 2>&1 | tail -1
