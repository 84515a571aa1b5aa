#!/bin/bash
# Instruction budget of k_fast_cells by phase: VALU / SALU / LDS instruction counters and time alone of builds that skip phase A, B
# (and everything after it) or C (-DFAST_SKIP_A/B/C: wrong results by design, counting only).   tools/fast_budget.sh
# Build the three libraries first (in the build container):
#   for v in A B C; do SD_OUT=tools/build/libsdslam_hip_skip$v.so SD_EXTRA_FLAGS=-DFAST_SKIP_$v python -m sdslam_amd.build --force; done
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for v in full skipC skipB skipA; do
  lib=$R/sdslam_amd/libsdslam_hip.so; [ $v != full ] && lib=$R/tools/build/libsdslam_hip_$v.so
  rm -rf $R/gpurun_out/fb_$v
  SD_LIB=$lib AMD_SERIALIZE_KERNEL=3 timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/fb_$v -o p -- python $R/bench.py --orb-only --steps 3 --warmup 1 --no-cpu-baseline --no-extras --unique 8 > $R/gpurun_out/fb_$v.log 2>&1
  SD_LIB=$lib AMD_SERIALIZE_KERNEL=3 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fbt_$v -o t -- python $R/bench.py --orb-only --steps 10 --warmup 2 --no-cpu-baseline --no-extras --unique 8 > $R/gpurun_out/fbt_$v.log 2>&1
  python - <<PY
import csv,glob,collections
d=collections.defaultdict(lambda:[0,0.0])
for f in glob.glob("$R/gpurun_out/fb_$v/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fast_cells" in r["Kernel_Name"]:
            d[r["Counter_Name"]][0]+=1; d[r["Counter_Name"]][1]+=float(r["Counter_Value"])
t=[0,0.0]
for f in glob.glob("$R/gpurun_out/fbt_$v/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fast_cells" in r["Kernel_Name"]: t[0]+=1; t[1]+=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
print("$v", {k:"%.1f M/launch"%(v[1]/v[0]/1e6) for k,v in d.items()}, "alone %.1f us/launch"%(t[1]/max(t[0],1)/1e3))
PY
done
