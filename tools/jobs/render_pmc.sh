#!/bin/bash
# frame kernel alone (tools/render_time.py): VALU / SALU / LDS instruction counts and busy cycles per launch (rocprofv3 --pmc with the kernel trace only)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/render_pmc
rm -rf $OUT; mkdir -p $OUT
( cd $REPO && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d $OUT -o p -- python tools/render_time.py ${RENDER_N:-32768} ${RENDER_TASK:-stack} > $OUT/log.txt 2>&1 )
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for fn in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "render_obs" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: f"{sum(v)/len(v):.4g}" for k, v in sorted(acc.items())}, "launches", max((len(v) for v in acc.values()), default=0))
PY
grep -h "render " $OUT/log.txt | tail -1
