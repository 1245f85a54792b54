#!/bin/bash
# same-box A/B of two builds of the library: gym_lowcostrobot_amd/liblcr_hip_ab.so (A, LCR_LIB_PATH) against liblcr_hip.so (B); arguments: tasks for tools/quick_times.py
for i in 1 2; do
  for t in "$@"; do
    echo "A: $(LCR_LIB_PATH=$PWD/gym_lowcostrobot_amd/liblcr_hip_ab.so python tools/quick_times.py $t --steps 100 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-20,100-170)"
    echo "B: $(python tools/quick_times.py $t --steps 100 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-20,100-170)"
  done
done
