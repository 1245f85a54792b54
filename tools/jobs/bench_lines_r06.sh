#!/bin/bash
# GPU job (round 6): the bench lines of BASELINE configs 2 (headline), 3, 4, 5 under the shipped default, with profiles/traffic.json and the PMC / ISA-mix files of the same kernel hash in place
mkdir -p gpurun_out
python bench.py > gpurun_out/bl_r06.json 2> gpurun_out/bl_err.txt
for c in 3 4 5; do python bench.py --config $c > gpurun_out/bl_r06_c$c.json 2>> gpurun_out/bl_err.txt; done
for f in gpurun_out/bl_r06*.json; do head -c 300 $f; echo; done
