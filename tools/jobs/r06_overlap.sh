#!/bin/bash
# GPU job (round 6): frames on a second stream (lcr_step) -- the tests that read frames, then BASELINE config 5 with and without the overlap
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -p no:cacheprovider -n 4 -k "image or render or frame or record or terminal or obs or facade or vecenv" 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed" | cut -c1-200 | tail -12 | tee gpurun_out/r06_overlap.txt
for ov in 0 1; do
  echo "LCR_RENDER_OVERLAP=$ov: $(LCR_RENDER_OVERLAP=$ov python bench.py --config 5 --no-cpu-baseline 2>/dev/null | tail -1 | head -c 260)" | tee -a gpurun_out/r06_overlap.txt
done
for ov in 0 1; do
  echo "LCR_RENDER_OVERLAP=$ov ReachCube + frames, 32 768 envs: $(LCR_RENDER_OVERLAP=$ov python bench.py --workload ReachCube-v0 --envs-per-gpu 32768 --obs both --no-cpu-baseline 2>/dev/null | tail -1 | head -c 260)" | tee -a gpurun_out/r06_overlap.txt
done
