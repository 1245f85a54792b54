#!/bin/bash
# Run ON THE GPU BOX: ms per step of one workload over batch sizes (one wave per SIMD => 65 536 envs fill the GPU once).
# Usage: tools/batch_size_sweep.sh [workload]
cd ${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-ReachCube-v0}
for n in 16384 32768 49152 65536 81920 98304 131072 262144; do
  echo -n "$n "; python bench.py --workload $WL --envs-per-gpu $n --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
