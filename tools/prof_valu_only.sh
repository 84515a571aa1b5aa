#!/bin/bash
# VALU / LDS instruction counts + durations of the extraction kernels only: bash tools/prof_valu_only.sh <tag>
set -e -o pipefail
TAG=${1:-v}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profv_$TAG
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -o p -- python $ROOT/bench.py --orb-only --steps 3 --warmup 1 --no-cpu-baseline --unique 8 > $OUT/sq.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/bench.py --orb-only --steps 10 --warmup 2 --no-cpu-baseline --unique 8 > $OUT/trace.log 2>&1
python $ROOT/tools/pmc_summary.py $OUT/sq $OUT/trace | python -c "
import json,sys
d=json.load(sys.stdin)
for k,e in d.items():
    if 'SQ_INSTS_VALU_per_launch' in e and e.get('duration_ns_samples',0)>0:
        print(k[:24], 'VALU/launch', round(e['SQ_INSTS_VALU_per_launch']/1e6,1),'M  LDS', round(e.get('SQ_INSTS_LDS_per_launch',0)/1e6,1),'M  SALU', round(e.get('SQ_INSTS_SALU_per_launch',0)/1e6,1), 'M  dur/launch us', round(e['duration_ns_per_launch']/1e3,1), 'launches', e['duration_ns_samples'])
"
