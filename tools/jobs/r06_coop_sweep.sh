#!/bin/bash
# GPU job (round 6): step time against LCR_COOP_MAX (coupled envs a wave solves cooperatively before it falls back to the 12-dim SIMT copy), then the per-wave phases
mkdir -p gpurun_out
for m in 0 1 2 3 4 6 8 12; do
  for t in reach push pick_place_ee; do
    echo "coop_max=$m $(LCR_COOP_MAX=$m python tools/quick_times.py $t --steps 100 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-20,100-170)"
  done
done 2>&1 | tee gpurun_out/coop_sweep.txt
python tools/newton_phases.py reach push lift 2>&1 | grep -v amdgpu.ids | tee gpurun_out/newton_phases.txt
