#!/bin/bash
# Run ON THE GPU BOX: SQ counters of the image-observation bench (render kernel).  Usage: tools/pmc_render.sh <tag>
TAG=${1:-img}; REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --workload StackTwoCubes-v0 --obs both --envs-per-gpu 32768 --steps 6 --warmup 2 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/a -o p -- $BENCH > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/b -o p -- $BENCH > $OUT/b.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("a", "b"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            if "render_obs" in k or "step_kernel" in k:
                print(k); [print("   ", c, x) for c, x in sorted(v.items())]
PY
