#!/bin/bash
# Alternating bench runs (full step and ORB-only) of several builds of the library: tools/ab_libs.sh N name=path ... ("head" = the in-tree build)
set -e
N=$1; shift
for k in $(seq 1 $N); do
  for nv in "$@"; do
    name=${nv%%=*}; path=${nv#*=}
    if [ "$path" = "head" ]; then unset SD_LIB; else export SD_LIB=$GRAFT_REPO_ROOT/$path; fi
    python bench.py --no-cpu-baseline --no-extras --steps 100 > gpurun_out/abl_${name}_$k.json
    python bench.py --no-cpu-baseline --no-extras --steps 100 --orb-only > gpurun_out/abl_${name}_orb$k.json
  done
done
python - "$@" <<P
import json,glob,sys
for nv in sys.argv[1:]:
    n=nv.split("=")[0]
    a=[json.load(open(f))["value"]/1e3 for f in sorted(glob.glob("gpurun_out/abl_%s_[0-9].json"%n))]
    o=[json.load(open(f))["value"]/1e3 for f in sorted(glob.glob("gpurun_out/abl_%s_orb[0-9].json"%n))]
    print(n, ["%.1f"%x for x in a], ["%.1f"%x for x in o])
P
