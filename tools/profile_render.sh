#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel trace + byte counters of the image-observation workload (config-5 shape).
# Usage: tools/profile_render.sh <tag>   -> gpurun_out/prof_<tag>_img/
TAG=${1:-r01}; REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out/prof_${TAG}_img; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --workload StackTwoCubes-v0 --obs both --envs-per-gpu 32768 --steps 20 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/bench_line.json 2> $OUT/trace.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o p -- $BENCH > $OUT/pmc_$C.log 2>&1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_SQ -o p -- $BENCH > $OUT/pmc_SQ.log 2>&1
python3 - <<PY
import csv, glob, collections, json
res = {}
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "render_obs" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for c in acc: res[c] = {"sum": acc[c], "rows": n[c]}
json.dump(res, open("$OUT/render_pmc_raw.json", "w"), indent=1)
print(json.dumps(res))
PY
cat $OUT/trace/*kernel_stats.csv | head -5; tail -1 $OUT/bench_line.json
