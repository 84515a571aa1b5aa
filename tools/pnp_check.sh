#!/bin/bash
# k_pnp after a change: parity (the RANSAC scenarios + EPnP + golden), WRITE_SIZE per launch, time alone / in the pipeline, stress legs.
#   tools/pnp_check.sh TAG
set -e -o pipefail
TAG=${1:-pnp}
python -m pytest tests/test_pnp_ransac.py tests/test_track_gpu.py -q -m gpu -x --no-header -p no:cacheprovider -k "pnp or epnp" > gpurun_out/${TAG}_t.log 2>&1 || { tail -30 gpurun_out/${TAG}_t.log; exit 1; }
tail -1 gpurun_out/${TAG}_t.log
bash tools/prof_write_only.sh $TAG
bash tools/trace_quick.sh $TAG | grep -i 'pnp\|sum\|frames' 
python bench.py --no-cpu-baseline --steps 100 > gpurun_out/${TAG}_bench.json
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench.json"))
print("full step %.1f k"%(d["value"]/1e3), {k:(round(v.get("ms_per_launch",v.get("ms_per_step",0)),2)) for k,v in d["stress"].items()})
PY
