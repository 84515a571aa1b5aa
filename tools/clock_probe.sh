#!/bin/bash
# Sample the shader clock and the package power while bench.py runs (GPU box): does the part hold 2.4 GHz under this load?
#   bash tools/clock_probe.sh [bench args]   -> gpurun_out/clock_probe.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/clock_probe.txt
: > $OUT
( for i in $(seq 1 400); do echo "t=$(date +%s.%N)" >> $OUT; rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -i "sclk\|power\|busy\|mclk" >> $OUT; sleep 0.25; done ) &
PROBE=$!
python $ROOT/bench.py --no-cpu-baseline --no-extras --steps 1500 "$@" > $ROOT/gpurun_out/clock_probe_bench.json 2>/dev/null
kill $PROBE 2>/dev/null
wait $PROBE 2>/dev/null
python - <<P
import re
t = open("$OUT").read()
s = [int(x) for x in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", t)] or [int(x) for x in re.findall(r"sclk[^\n]*?\((\d+)Mhz\)", t)]
p = [float(x) for x in re.findall(r"Power \(W\): ([\d.]+)", t)]
print("sclk samples", len(s), "min/median/max", (min(s), sorted(s)[len(s)//2], max(s)) if s else None)
print("power samples", len(p), "min/median/max", (min(p), sorted(p)[len(p)//2], max(p)) if p else None)
P
tail -c 300 $ROOT/gpurun_out/clock_probe_bench.json | head -c 300; echo
