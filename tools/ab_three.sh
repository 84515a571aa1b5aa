set -e
python -m pytest tests/test_orb_gpu.py tests/test_golden.py -q -m gpu -x --no-header -p no:cacheprovider > gpurun_out/r3three_t.log 2>&1 || { tail -30 gpurun_out/r3three_t.log; exit 1; }
tail -1 gpurun_out/r3three_t.log
NC=$GRAFT_REPO_ROOT/tools/build/libsdslam_hip_nc.so; BASE=$GRAFT_REPO_ROOT/tools/build/libsdslam_hip_base.so
for k in 1 2 3; do
  for v in full base; do
    if [ $v = full ]; then unset SD_LIB; elif [ $v = nc ]; then export SD_LIB=$NC; else export SD_LIB=$BASE; fi
    python bench.py --no-cpu-baseline --no-extras --steps 100 > gpurun_out/r3three_${v}_$k.json
    python bench.py --no-cpu-baseline --no-extras --steps 100 --orb-only > gpurun_out/r3three_${v}_orb$k.json
  done
done
python - <<P
import json,glob
for v in ("full","base"):
    a=[json.load(open(f))["value"]/1e3 for f in sorted(glob.glob("gpurun_out/r3three_%s_[123].json"%v))]
    o=[json.load(open(f))["value"]/1e3 for f in sorted(glob.glob("gpurun_out/r3three_%s_orb[123].json"%v))]
    print(v, ["%.1f"%x for x in a], ["%.1f"%x for x in o])
P
unset SD_LIB; bash tools/trace_quick.sh full --orb-only | grep "k_pyr\|k_fast\|k_orient\|k_blur"
