#!/bin/bash
# GPU job: what the Newton kernel's time is made of (iteration cap, line-search cap, job size).   bash tools/jobs/newton_timing.sh
mkdir -p gpurun_out
out=gpurun_out/nt_iters.log
: > $out
for it in 1 2 3 4 6 10 20; do python tools/quick_times.py reach --newton-iters $it --steps 100; done >> $out 2>&1
for ls in 1 2 4 8; do python tools/quick_times.py reach --ls-iters $ls --steps 100; done >> $out 2>&1
for n in 16384 32768 49152 65536 98304 131072; do python tools/quick_times.py reach --n $n --steps 100; done >> $out 2>&1
python tools/quick_times.py reach --newton-tol 1e-4 --ls-tol 1e-2 --steps 100 >> $out 2>&1
grep -v amdgpu.ids $out
