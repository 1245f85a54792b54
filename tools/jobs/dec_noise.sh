#!/bin/bash
# GPU job (experiment): the decrement's rounding floor -- step time and parity for a few values
mkdir -p gpurun_out
for v in 0 1e-13 1e-12 1e-11; do
  echo "== LCR_DEC_NOISE2=$v"
  LCR_DEC_NOISE2=$v python tools/quick_times.py reach push --steps 100 2>&1 | grep -v amdgpu.ids
  LCR_DEC_NOISE2=$v python tools/newton_dev_check.py reach,lift,stack 4096 5 2>&1 | grep -v amdgpu.ids
done > gpurun_out/dec_noise.log 2>&1
cat gpurun_out/dec_noise.log
