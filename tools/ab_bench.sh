#!/bin/bash
# A/B of two bench.py configurations by ALTERNATING runs (A B A B ...), one JSON line per run into gpurun_out/<tag>_{a,b}<k>.json.
#   tools/ab_bench.sh TAG N "ARGS_A" "ARGS_B" ["ENV_B"]      ENV_B e.g. SD_LIB=tools/build/libsdslam_hip_base.so (another build of the library)
# (decisions on the full step need alternating runs; a single pair is inside the run-to-run noise)
set -e
tag=$1; n=$2; a=$3; b=$4; envb=$5
mkdir -p gpurun_out
for k in $(seq 1 $n); do
  python bench.py --no-cpu-baseline --no-extras --steps 100 $a > gpurun_out/${tag}_a$k.json
  env $envb python bench.py --no-cpu-baseline --no-extras --steps 100 $b > gpurun_out/${tag}_b$k.json
done
python - <<PY
import json,glob
for side,args in (("a","$a"),("b","$b")):
    v=[json.load(open(f)) for f in sorted(glob.glob("gpurun_out/${tag}_%s*.json"%side))]
    print(side, repr(args), ["%.1f k"%(x["value"]/1e3) for x in v], "fast %.2f ms"%(sum(x["stages_ms_per_step"]["fast_nms"] for x in v)/len(v)),
          {k:round(sum(x["stages_ms_per_step"][k] for x in v)/len(v),2) for k in v[0]["stages_ms_per_step"]})
PY
