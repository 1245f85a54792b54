#!/bin/bash
# GPU job (experiment): kernel vs oracle Newton iteration counts under a few solver settings
for kw in "" "ls_iters=16,ls_tol=1e-4" "ls_iters=16,ls_tol=1e-6" "newton_tol=1e-5" "ls_iters=1"; do
  echo "== DEV_KW=$kw"
  DEV_KW=$kw DEV_DUMP=1 python tools/newton_dev_check.py reach 2048 4 2>&1 | grep -v amdgpu.ids | tail -4
done
