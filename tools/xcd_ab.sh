set -e
python -m pytest tests/test_orb_gpu.py -q -m gpu -x --no-header -p no:cacheprovider > gpurun_out/xcd_t.log 2>&1 || { tail -20 gpurun_out/xcd_t.log; exit 1; }
tail -2 gpurun_out/xcd_t.log
bash tools/ab_bench.sh xcd 3 "" "" "SD_LIB=tools/build/libsdslam_hip_base.so"
bash tools/ab_bench.sh xcdorb 2 "--orb-only" "--orb-only" "SD_LIB=tools/build/libsdslam_hip_base.so"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/xcd_fetch -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --unique 8 > $R/gpurun_out/xcd_fetch.log 2>&1
python $R/tools/fetch_calib_fast.py $R/gpurun_out/xcd_fetch 1024 > $R/gpurun_out/xcd_fetch_calib.json
python - <<PY
import csv,glob,collections
d=collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/xcd_fetch/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"])*1024)
for k,v in d.items(): print(k, len(v), "raw MB/launch %.1f"%(sum(v)/len(v)/1e6))
PY
