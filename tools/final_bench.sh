set -e
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
echo default done
python bench.py --no-extras --no-cpu-baseline --pose-solver poseopt > gpurun_out/r02_bench_poseopt.json 2>/dev/null
python bench.py --no-extras --no-cpu-baseline --pose-solver motion_model > gpurun_out/r02_bench_motion_model.json 2>/dev/null
python bench.py --no-extras --no-cpu-baseline --pose-solver track > gpurun_out/r02_bench_track.json 2>/dev/null
echo solvers done
python bench.py --no-extras --no-cpu-baseline --hamming > gpurun_out/r02_bench_hamming.json 2>/dev/null
python bench.py --no-extras --no-cpu-baseline --orb-only > gpurun_out/r02_bench_orb.json 2>/dev/null
python bench.py --no-extras --no-cpu-baseline --res 1280x720 --batch 1024 > gpurun_out/r02_bench_720.json 2>/dev/null
python bench.py --no-extras --no-cpu-baseline --batch 1 --pose-solver motion_model --steps 2000 > gpurun_out/r02_bench_b1.json 2>/dev/null
echo variants done
python tools/bench_reloc.py 1024 > gpurun_out/r02_bench_reloc.txt 2>&1 || true
python - <<P
import json,glob
for f in sorted(glob.glob("gpurun_out/r02_bench_*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], round(d["value"],1), round(d["ms_per_step"],3))
P
tail -5 gpurun_out/r02_bench_reloc.txt
