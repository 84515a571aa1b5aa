#!/bin/bash
# A change in the library against the previous build: parity of the extractor tests, then alternating bench runs (full step and
# ORB-only), then kernel times in the pipeline / alone.   tools/ab_lib.sh TAG [pytest -k expression]
#   baseline = tools/build/libsdslam_hip_base.so (built from the previous commit), candidate = sdslam_amd/libsdslam_hip.so
set -e -o pipefail
TAG=$1; K=${2:-}
python -m pytest tests/test_orb_gpu.py tests/test_golden.py -q -m gpu -x --no-header -p no:cacheprovider ${K:+-k "$K"} > gpurun_out/${TAG}_t.log 2>&1 || { tail -30 gpurun_out/${TAG}_t.log; exit 1; }
tail -1 gpurun_out/${TAG}_t.log
bash tools/ab_bench.sh ${TAG} 3 "" "" "SD_LIB=tools/build/libsdslam_hip_base.so"
bash tools/ab_bench.sh ${TAG}orb 2 "--orb-only" "--orb-only" "SD_LIB=tools/build/libsdslam_hip_base.so"
bash tools/trace_quick.sh ${TAG} > gpurun_out/${TAG}_tq.txt; head -14 gpurun_out/${TAG}_tq.txt
