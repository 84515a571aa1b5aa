#!/bin/bash
# LDS-side counters of the extraction kernels (one PMC pass, orb-only, 3 steps): bank conflicts, LDS busy cycles, waits.
#   bash tools/prof_lds.sh <tag>     (on the GPU box, through gpurun)
set -e -o pipefail
TAG=${1:-lds}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras --unique 8 --orb-only"
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/lds -o p -- python $ROOT/bench.py $ARGS > $OUT/lds.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/act -o p -- python $ROOT/bench.py $ARGS > $OUT/act.log 2>&1
python - <<P
import csv, glob, collections
for sub in ("lds", "act"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    for k in sorted(acc):
        n = len(cnt[k])
        print(sub, k, n, {c: round(v / n) for c, v in sorted(acc[k].items())})
P
