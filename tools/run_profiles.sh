#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + three PMC passes of the default bench workload.
#   bash tools/run_profiles.sh r01e
# Outputs under gpurun_out/prof_<tag>/{trace,sq,fetch,write}; summarise with tools/pmc_summary.py.
set -e -o pipefail
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras --unique 8"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/bench.py --no-cpu-baseline --no-extras > $OUT/trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/sq -o p -- python $ROOT/bench.py $ARGS > $OUT/sq.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python $ROOT/bench.py $ARGS > $OUT/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python $ROOT/bench.py $ARGS > $OUT/write.log 2>&1
python $ROOT/tools/pmc_summary.py $OUT/sq $OUT/fetch $OUT/write $OUT/trace > $OUT/pmc_summary.json
python $ROOT/tools/fetch_calib_fast.py $OUT/fetch 1024 > $OUT/fetch_calibration_fast.json
AMD_SERIALIZE_KERNEL=3 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serialized -o t -- python $ROOT/bench.py --no-cpu-baseline --no-extras --steps 30 > $OUT/trace_serialized.log 2>&1
ls $OUT
