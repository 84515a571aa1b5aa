#!/bin/bash
# WRITE_SIZE pass only (scratch spills show up here): bash tools/prof_write_only.sh <tag>   [env: SD_ALIGN_MINW]
set -e -o pipefail
TAG=${1:-w}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profw_$TAG
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --unique 8 > $OUT/write.log 2>&1
python $ROOT/tools/pmc_summary.py $OUT/write | python -c "
import json,sys
d=json.load(sys.stdin)
for k,e in d.items():
    if 'WRITE_bytes_per_launch' in e and ('align' in k or 'pnp' in k): print(k, round(e['WRITE_bytes_per_launch']/1e6,1), 'MB/launch')
"
