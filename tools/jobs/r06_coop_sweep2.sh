#!/bin/bash
# GPU job (round 6): step time against LCR_COOP_MAX with four patients per pass (one-cube tasks and StackTwoCubes' arm + cube patients), then PushCubeLoop's phases
mkdir -p gpurun_out
for m in 4 8 12 16 24 32 64; do
  for t in push pick_place_ee stack; do
    echo "coop_max=$m $(LCR_COOP_MAX=$m python tools/quick_times.py $t --steps 100 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-20,100-170)"
  done
  echo "coop_max=$m $(LCR_COOP_MAX=$m python tools/quick_times.py stack --n 32768 --steps 100 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-20,100-170)"
done 2>&1 | tee gpurun_out/r06_coop_sweep2.txt
python tools/newton_phases.py push_loop 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_loop_phases.txt
