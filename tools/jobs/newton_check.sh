#!/bin/bash
# GPU job: Newton kernels -- step times of every task at 65 536 envs, re-synchronised parity against the oracle, a few sizes of Reach
mkdir -p gpurun_out
python tools/quick_times.py --steps 100 2>&1 | grep -v amdgpu.ids > gpurun_out/nt_times.log
for n in 16384 49152; do python tools/quick_times.py reach --n $n --steps 100; done 2>&1 | grep -v amdgpu.ids >> gpurun_out/nt_times.log
python tools/newton_dev_check.py reach,push,lift,pick_place,stack,push_loop 4096 6 2>&1 | grep -v amdgpu.ids > gpurun_out/nt_parity.log
cat gpurun_out/nt_times.log gpurun_out/nt_parity.log
