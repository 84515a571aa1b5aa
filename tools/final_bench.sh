# Every bench variant of a round, one JSON line each under gpurun_out/<TAG>_bench_*.json (copy the ones to be judged to profiles/).
#   bash tools/final_bench.sh r03
set -e
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
echo default done
Q="--no-extras --no-cpu-baseline"
python bench.py $Q --pyramid 5x2.0 > gpurun_out/${TAG}_bench_p5.json 2>/dev/null
python bench.py $Q --pose-solver poseopt > gpurun_out/${TAG}_bench_poseopt.json 2>/dev/null
python bench.py $Q --pose-solver motion_model > gpurun_out/${TAG}_bench_motion_model.json 2>/dev/null
python bench.py $Q --pose-solver motion_model --pyramid 5x2.0 > gpurun_out/${TAG}_bench_motion_model_p5.json 2>/dev/null
python bench.py $Q --pose-solver track > gpurun_out/${TAG}_bench_track.json 2>/dev/null
echo solvers done
python bench.py $Q --hamming > gpurun_out/${TAG}_bench_hamming.json 2>/dev/null
python bench.py $Q --orb-only > gpurun_out/${TAG}_bench_orb.json 2>/dev/null
python bench.py $Q --orb-only --pyramid 5x2.0 > gpurun_out/${TAG}_bench_orb_p5.json 2>/dev/null
python bench.py $Q --res 1280x720 --batch 1024 > gpurun_out/${TAG}_bench_720.json 2>/dev/null
python bench.py $Q --batch 1 --pose-solver motion_model --steps 2000 > gpurun_out/${TAG}_bench_b1.json 2>/dev/null
echo variants done
python tools/bench_reloc.py 1024 > gpurun_out/${TAG}_bench_reloc.txt 2>&1 || true
python - <<P
import json,glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench_*.json")):
    d=json.loads([l for l in open(f) if l.startswith("{")][-1]); print(f.split("/")[-1], round(d["value"],1), round(d["ms_per_step"],3))
P
tail -3 gpurun_out/${TAG}_bench_reloc.txt
